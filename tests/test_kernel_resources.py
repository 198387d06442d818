"""Register / scratch / LDS budgets of the shipped kernels, read from the metadata of the built library (CPU tier: hipcc
cross-compiles, nothing runs).  The transforms are VALU-bound with one workgroup per CU at N = 32768: a spill in their hot
loops or an inner product that drops below two waves per SIMD is a performance regression that no parity test notices."""
import os
import re
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(
    not (os.path.exists("/opt/rocm/lib/llvm/bin/clang-offload-bundler") and shutil.which("objcopy") and shutil.which("c++filt")),
    reason="needs the ROCm LLVM tools")


@pytest.fixture(scope="module")
def meta():
    from hehub_amd.build import build_lib
    from kernel_meta import kernel_meta
    return kernel_meta(build_lib())


def pick(meta, pattern):
    found = {k: v for k, v in meta.items() if re.search(pattern, k)}
    assert found, pattern
    return found


def test_every_tiled_size_is_built(meta):
    for logn in range(11, 16):
        pick(meta, rf"k_ntt_fwd<{logn}>")
        pick(meta, rf"k_ntt_inv<{logn}, ")
        for flav in range(6):
            pick(meta, rf"k_ntt_fwd_drop<{logn}, {flav}[,>]")


def test_forward_transforms_do_not_spill(meta):
    for name, r in pick(meta, r"k_ntt_fwd<\d+>").items():
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, (name, r)
        assert r["vgpr_count"] <= 128, (name, r)           # four waves per SIMD


def test_n32768_kernels(meta):
    """one workgroup (16 waves x <= 128 VGPRs, 144-148 KiB of LDS) per CU: no spills in the C3 kernels"""
    for name, r in pick(meta, r"k_ntt_(fwd|inv)<15").items():
        assert r["vgpr_spill_count"] == 0, (name, r)
        assert 140 * 1024 <= r["group_segment_fixed_size"] <= 160 * 1024, (name, r)
    for name, r in pick(meta, r"k_ntt_fwd_drop<15, [1-5][,>]").items():
        assert r["vgpr_spill_count"] <= 2, (name, r)


def test_fused_drop_flavours(meta):
    """compile-time flavours (the CKKS / BGV pipelines, rotations): at most two spilled registers; the run-time flavour 0 (rotations,
    hybrid key switch inputs) is known to spill ~40-48 and is not on the headline path"""
    for name, r in pick(meta, r"k_ntt_fwd_drop<\d+, [1-5][,>]").items():
        assert r["vgpr_spill_count"] <= 2, (name, r)
    for name, r in pick(meta, r"k_ntt_fwd_drop<\d+, 0[,>]").items():
        assert r["vgpr_spill_count"] <= 48, (name, r)


def test_inverse_small_sizes_budget(meta):
    """multi-limb workgroups at N <= 8192, one limb at 16384: a few spilled registers are the measured optimum (an occupancy of
    four waves per SIMD matters more); the budget is today's count"""
    for name, r in pick(meta, r"k_ntt_inv<1[1-4], ").items():
        assert r["vgpr_spill_count"] <= 24, (name, r)


def test_inner_product_occupancy(meta):
    # (<PT, false>: the kernel of parity level B and of 48-bit rows; <PT, true>: level A with 40-bit rows, its own instantiation)
    for p40 in ("false", "true"):
        (name, r), = pick(meta, rf"k_ks_inner_blk<4, {p40}>").items()
        assert r["vgpr_spill_count"] == 0 and r["vgpr_count"] <= 256, (name, r)     # two waves per SIMD
        (name, r), = pick(meta, rf"k_ks_inner_blk<2, {p40}>").items()
        assert r["vgpr_spill_count"] == 0 and r["vgpr_count"] <= 128, (name, r)     # four


def test_no_kernel_spills_scalars_or_exceeds_lds(meta):
    for name, r in meta.items():
        if "k_hks_" not in name:   # (the hybrid extension's conversions with >= 5 special primes keep their constants in spilled SGPRs)
            assert r.get("sgpr_spill_count", 0) == 0, (name, r)
        assert r.get("group_segment_fixed_size", 0) <= 160 * 1024, (name, r)
    for name, r in pick(meta, r"^(void )?k_(tensor|poly_binary|poly_unary|drop_rem|drop_fin|gather|vec)").items():
        assert r["vgpr_spill_count"] == 0, (name, r)


def test_no_wide_buffer_store_with_scalar_offset():
    """gfx950: a buffer store of more than 8 bytes per lane may still be reading its data registers when the next instruction
    issues.  The assembler pads that only when the store has NO scalar offset register; with one, a VALU instruction right
    behind the store that overwrites a data register changes what the last lanes of each 16-lane pass write (found in round 2:
    rare wrong words in rotations, NOTES.md).  So no shipped kernel may contain such a store."""
    from hehub_amd.build import build_lib
    from kernel_meta import disassembly
    text = disassembly(build_lib())
    stores = [l for l in text.splitlines() if re.search(r"buffer_store_(dwordx[34]|format_xyzw?)\b", l)]
    assert stores                                      # the fused drop's epilogue uses them
    bad = [l for l in stores if re.search(r"buffer_store_\w+\s+v\[\d+:\d+\],\s*(v\d+|off),\s*s\[\d+:\d+\],\s*s\d+", l)]
    assert not bad, bad[:3]


def test_vcc_carry_wait_states():
    """The hand-scheduled product chains (hp_device.h: hp_harvey_lazy_nq, hp_barrett_lazy_nq, hp_butterfly2_nq*) read the carry
    that a v_mad_u64_u32 left in vcc with `v_addc_co_u32 v, vcc, 0, 0, vcc`.  The spacing the compiler itself keeps between the
    two on gfx950 is two independent instructions (or `s_nop 1`); the asm strings place them by hand, so the shipped code is
    checked: every such reader has its vcc writer at least three instructions above it, with no label (branch target) in
    between."""
    from hehub_amd.build import build_lib
    from kernel_meta import disassembly
    lines = [l.split("//")[0].rstrip() for l in disassembly(build_lib()).splitlines()]
    lines = [l for l in lines if l.strip()]
    readers = [i for i, l in enumerate(lines) if re.search(r"v_addc_co_u32_e64 v\d+, vcc, 0, 0, vcc", l)]
    assert len(readers) > 10000          # 2 per dual butterfly, 240 butterflies per tiled kernel
    writes_vcc = re.compile(r"^\s*(v_\w+\s+(v\[\d+:\d+\]|v\d+),\s*vcc\b|v_cmp\w*\s+vcc\b|s_\w+\s+vcc\b)")
    bad = []
    for i in readers:
        dist = None
        for back in range(1, 12):
            l = lines[i - back]
            if l.rstrip().endswith(":") or not l.startswith(("\t", " ")):      # a label: cannot vouch for the path into it
                break
            m = re.match(r"\s*s_nop (\d+)", l)
            if m and int(m.group(1)) >= 1 and back == 1:
                dist = 99                                                       # explicit wait states right in front of the reader
                break
            if writes_vcc.match(l):
                dist = back
                break
        if dist is None or dist < 3:
            bad.append((i, dist, lines[max(0, i - 4):i + 1]))
    assert not bad, bad[:2]


def test_level_a_kernels(meta):
    """hp_ntt_a.hip (FP64 residue butterflies): every tiled size and flavour is built, four waves per SIMD, no spills in anything
    the C3 pipeline and the single drops launch (the BGV-with-addend flavour, the two-drops flavours below N = 32768 and the smallest
    inverse keep 2-4 spilled registers)"""
    for logn in range(11, 16):
        for spread in ("false", "true"):
            pick(meta, rf"k_ntt_fwd_a<{logn}, {spread}>")
        for ps in ("false", "true"):
            pick(meta, rf"k_ntt_inv_a<{logn}, {ps}>")
        for flav in range(1, 8):
            pick(meta, rf"k_ntt_fwd_drop_a<{logn}, {flav}>")
    for name, r in pick(meta, r"k_ntt_(fwd|inv|fwd_drop)_a<").items():
        assert r["vgpr_count"] <= 128 and r["sgpr_spill_count"] == 0, (name, r)
        assert r["vgpr_spill_count"] <= 4, (name, r)
    for name, r in pick(meta, r"k_ntt_fwd_a<\d+, |k_ntt_inv_a<1[2-5], |k_ntt_fwd_drop_a<\d+, [1235]>|k_ntt_fwd_drop_a<15, [67]>").items():
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0, (name, r)
    for name, r in pick(meta, r"k_ntt_(fwd|inv|fwd_drop)_a<15").items():
        assert 140 * 1024 <= r["group_segment_fixed_size"] <= 160 * 1024, (name, r)


def test_level_a_butterfly_is_not_contracted():
    """the error-free product needs h = RN(x w) and l = fma(x, w, -h) as TWO roundings: the shipped code object of hp_ntt_a.hip must hold
    as many v_mul_f64 as v_fma_f64-pairs of the butterfly (a contracted build would have turned multiplies into FMAs)"""
    from kernel_meta import disassembly

    text = disassembly()
    body = text[text.index("k_ntt_fwd_aILi15E"):]
    body = body[:body.index("s_endpgm")]
    mul, fma, rnd = body.count("v_mul_f64"), body.count("v_fma_f64"), body.count("v_rndne_f64")
    # per butterfly: 2 multiplies (x w, x u), 2 fused multiply-adds, 1 rint; 240 butterflies, the same again per reduction / canonicalisation
    assert mul >= 480 and fma >= 480 and rnd >= 240, (mul, fma, rnd)
    assert abs(mul - fma) <= 8, (mul, fma)
