// hehub_amd_ext.hpp -- what the MI355X layer adds to hehub's public interface, in namespace hehub::amd.  Everything here is an
// ADDITION next to hehub's one-ciphertext-per-call functions (ckks.h:270-313, bgv.h:150-167), never a replacement:
//
//   * the engine handle, the parity level, PCIe accounting;
//   * BATCHED forms of the scheme-level calls: std::vector<CkksCt> in, one engine call, std::vector<CkksCt> out.  Element i of a
//     result is word for word what the single call of hehub's interface returns for element i (same checks, same exceptions --
//     a batch whose members do not all have one shape simply runs as that loop of single calls).  hehub's callers loop over
//     independent ciphertexts (src/circuits/linear_algebra.h:109-133, bench/benchmarks.cpp:24-35); one C3 ciphertext fills
//     10 .. 100 of the 256 CUs, a batch of 256 fills the GPU: 29.6 k instead of 4.3 k hom-mult/s;
//   * lanes: how many independent single calls may overlap on the device (own-mirror build; see layer.hpp "lanes").
//
// Include it after hehub_amd/host/hehub.hpp (which does so itself) or, in a build against hehub's OWN headers over the binding
// (-DHEHUB_AMD_BIND_REFERENCE), after "fhe/ckks/ckks.h", "fhe/bgv/bgv.h" and "fhe/primitives/keys.h".
#pragma once

#include <cstddef>
#include <vector>

struct hp_ctx;

namespace hehub {
namespace amd {

/// The process-wide engine (device 0 unless HEHUB_AMD_DEVICE is set); created on first use, like the
/// reference's lazily filled global caches (ntt.cpp:107-143).  Throws std::runtime_error when no GPU
/// or no engine library is available -- there is no CPU fallback.
hp_ctx *engine();
/// Parity level of the process-wide engine (include/hehub_amd.h: hp_ctx_set_parity_level; also HP_PARITY_LEVEL=A in the environment).
/// false = B (default): every word is hehub's raw lazy word.  true = A: ckks / bgv mult, relinearize, rotate, rescale, mod_switch return
/// the canonical residue of every word (reduce_strict of hehub's word; decryptions are identical) through the FP64 transforms -- 35 % more
/// hom-mult/s at N = 32768.  The NTT / mod-arith primitives and the operators are never affected.
void set_parity_level_a(bool on);
bool parity_level_a();

/// bytes and calls that crossed PCIe through this layer since the process started, and engine calls made
struct TransferStats {
    unsigned long long h2d_bytes = 0, d2h_bytes = 0, h2d_copies = 0, d2h_copies = 0, engine_calls = 0, host_blocks_registered = 0;
    // device copies thrown away because host words were handed out WRITABLE (operator[], components(), begin() / end() / last() on a
    // non-const vector): each one costs a re-upload at the next engine call (and a key-cache miss for a key polynomial).  Code that
    // only reads should go through view() / a const reference; this counter shows when it does not.
    unsigned long long device_copies_invalidated = 0;
    // device-side waits between lanes (one event each): how often a call depended on a call that ran on another lane
    unsigned long long lane_waits = 0;
    // deferred mode: recorded calls that have run, and the batched engine calls they ran as
    // (deferred_fused: ckks::mult + rescale_inplace / bgv mult + mod_switch_inplace triples that ran as the engine's one-call pipeline)
    unsigned long long deferred_calls = 0, deferred_groups = 0, deferred_fused = 0;
    // ... and the groups of rotations / conjugations that ran with a key per ciphertext (hp_dev_ckks_rotate_many)
    unsigned long long deferred_many_key_groups = 0;
    unsigned long long deferred_chain_sums = 0;   // ... += / -= chains of add / sub calls that ran as ONE pass (hp_dev_poly_fold_rows): calls folded
    // devices: engine calls per device rank, and the copies between ranks an operand on the wrong device cost (hp_memcpy_peer_async)
    unsigned long long calls_by_device[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long peer_copies = 0, peer_bytes = 0;
};
TransferStats transfer_stats();

/// Lanes: independent single-ciphertext calls are spread over this many streams of the engine and overlap on the device; a call
/// waits, on the device, for exactly the calls that produced its operands.  Default 4 (HEHUB_AMD_LANES), at most 8; 1 = everything on
/// one stream.  The binding build (hehub's own host-memory objects) always has one.  set_lanes() drains the device first.
/// A caller that looks at every result before its next call leaves nothing in flight: while every lane is known idle (after a look
/// or synchronize()) the layer stays on the lane it used last -- that lane's workspace is the one in the Infinity Cache -- and
/// spreads calls over the lanes only while something is running.  A look at ONE limb (`ct[1][0][i]`, `view(k)`) downloads that limb.
int lanes();
void set_lanes(int n);
/// Devices (own-mirror build): the GPUs of the node the layer spreads hehub's calls over.  hehub has no devices; its callers
/// hold many INDEPENDENT ciphertexts (src/circuits/linear_algebra.h:109-133, examples/ckks_example.cpp:10-27).  With n device ranks
///   * a call runs on the device where its first device-resident operand lives; a call on host-only operands goes to the next rank
///     round robin and uploads them there -- independent ciphertexts spread over the GPUs, a dependent chain stays on its GPU;
///   * recorded calls are grouped per rank; the groups of different ranks run side by side;
///   * the batched forms below cut a batch into parts, one per rank (no collective): an element stays where its operand lives, host-only
///     elements fill the ranks up to batch / n each, contiguously;
///   * keys and tables are replicated per rank on first use; an operand found on another rank is copied over one xGMI link
///     (TransferStats::peer_copies).
/// A vector has one device copy: an operand SHARED by calls on several ranks is moved, not replicated (give each rank its own copy, or keep
/// such a workload on one rank); keys are the exception (assembled per rank).
/// Results are word for word those of one device.  Default: one rank on HIP device HEHUB_AMD_DEVICE (0); HEHUB_AMD_DEVICES=<n> = devices
/// 0 .. n-1, HEHUB_AMD_DEVICES=<a>,<b>,.. = those devices (a device may repeat: ranks then share it -- how the one-GPU tests run this).
/// set_devices() drains the layer first; a rank that has been used keeps its device.  At most 8 ranks.
int devices();
void set_devices(int n);
void set_devices(const std::vector<int> &hip_devices);
/// Deferred mode (own-mirror build; ON by default since round 6 -- HEHUB_AMD_DEFER=0 in the environment or set_deferred(false) give
/// the call-by-call behaviour back; the binding build over hehub's own host-memory objects never defers): the scheme-level calls of hehub's
/// interface -- mult_low_level, relinearize, mult, rotate, conjugate, rescale_inplace, mod_switch_inplace, add / sub of ciphertexts,
/// polynomial products (mult_plain), += / -= and in-place transforms of polynomials, rns_base_transform from one modulus (the plaintext
/// lift of bgv::add_plain / sub_plain / mult_plain) and copies of results that are still pending --
/// are RECORDED with all their argument checks made and their result objects returned; they run when somebody needs words (a look at
/// a result, a call that cannot be recorded, synchronize(), 1024 recorded calls), grouped: recorded calls with one signature whose
/// operands are ready run as ONE batched engine call (rotations and conjugations group across keys and steps: the engine takes a key
/// per ciphertext; a chain `acc = add(acc, term)` whose intermediate sums nobody else holds runs as one pass over its terms).
/// An unchanged loop over independent ciphertexts thereby gets the batch rate
/// (layer.hpp "deferred execution", deferred_record.cpp / deferred_run.cpp).  Results are word for word those of the eager calls, and so
/// are the exceptions: what the engine would refuse (a modulus the transforms cannot use, an even modulus in a product, a rotation step out
/// of range) is refused by the recording call (hp_check_chain), like every argument check of hehub's; only a failure of the DEVICE (a HIP
/// error, out of memory) surfaces when the queue runs.  set_deferred(false) runs what is pending.
void set_deferred(bool on);
bool deferred();
/// Upload a vector's host words now (own-mirror build; both copies stay current).  hehub's objects are created on the host; the layer
/// uploads one when a call first reads it and keeps the device copy WITH THE OBJECT -- but a copy of a host-only object is a host copy.
/// An operand that is copied and then transformed call after call (the encoded diagonal inside ckks::mult_plain, ckks/arith.cpp:47-53)
/// crosses PCIe every time unless it is resident first.
void prefetch(const RnsIntVec &v);
/// wait until everything the layer has enqueued (or recorded) so far has run (hehub's interface has no such call: its functions are synchronous;
/// here a result is waited for when somebody looks at its words -- this is for timing loops)
void synchronize();

// ---- batched forms ---------------------------------------------------------------------------------------------------------
// ckks.h:270  mult(ct1, ct2, relin_key), element by element
std::vector<ckks::CkksCt> mult(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, const RlweKsk &relin_key);
// ckks.h:270 + :313  mult followed by rescale_inplace(ct, 1): the engine's fused pipeline (one tensor product, one key switch, the
// drop of the special prime and of q_last) -- the "hom-mult" of the headline metric
std::vector<ckks::CkksCt> mult_rescale(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b, const RlweKsk &relin_key);
// ckks.h:313  rescale_inplace(ct, 1) on every element
void rescale_inplace(std::vector<ckks::CkksCt> &cts);
// ckks.h:284 / :282  rotate(ct, rot_key, step) / conjugate(ct, conj_key), one key for the whole batch
std::vector<ckks::CkksCt> rotate(const std::vector<ckks::CkksCt> &cts, const RlweKsk &rot_key, size_t step);
std::vector<ckks::CkksCt> conjugate(const std::vector<ckks::CkksCt> &cts, const RlweKsk &conj_key);
// ckks.h:284  rotate(cts[i], *rot_keys[i], steps[i]): EVERY ciphertext under its own key and step, one engine call -- the rotations of
// one vector under the keys of a rotation key set (src/circuits/linear_algebra.h:123-130)
std::vector<ckks::CkksCt> rotate(const std::vector<ckks::CkksCt> &cts, const std::vector<const RlweKsk *> &rot_keys,
                                 const std::vector<size_t> &steps);
// ... and ONE ciphertext under many keys: element i = rotate(ct, *rot_keys[i], steps[i])
std::vector<ckks::CkksCt> rotate(const ckks::CkksCt &ct, const std::vector<const RlweKsk *> &rot_keys, const std::vector<size_t> &steps);
// ckks.h:73-89  add / sub, element by element
std::vector<ckks::CkksCt> add(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b);
std::vector<ckks::CkksCt> sub(const std::vector<ckks::CkksCt> &a, const std::vector<ckks::CkksCt> &b);
// bgv.h:150-159  mult_low_level + relinearize, element by element; _mod_switch: followed by mod_switch_inplace(ct, 1) (bgv.h:167)
std::vector<bgv::BgvCt> mult(const std::vector<bgv::BgvCt> &a, const std::vector<bgv::BgvCt> &b, const RlweKsk &relin_key);
std::vector<bgv::BgvCt> mult_mod_switch(const std::vector<bgv::BgvCt> &a, const std::vector<bgv::BgvCt> &b, const RlweKsk &relin_key);
void mod_switch_inplace(std::vector<bgv::BgvCt> &cts);

} // namespace amd
} // namespace hehub
