"""What bench.py prints: ONE compact contract line last, the detailed sections before it and in a side file.

The driver reads the LAST stdout line as JSON.  Round 5's line carried every section and had grown to 34.5 KB: the driver's
record of it was `parsed: null`.  So the last line is the contract (a few KB, bounded by LINE_LIMIT and checked before it is printed),
every detailed section goes

  * to `bench_sections.json` in the working directory (`HP_BENCH_SECTIONS=<path>` moves it, `HP_BENCH_SECTIONS=` = no file), and
  * to stdout BEFORE the contract line, one line per section, `#section <name> <json>` -- never starting with `{`; a BRIEF copy (floats to 5
    significant digits, long explanatory strings cut, objects nested deeper than three levels left to the side file) so that everything
    bench.py prints stays near 20 KB.

`collect(stdout)` puts the two back together (tests, tools).  No NaN / Infinity anywhere: non-finite floats become null.
"""
from __future__ import annotations

import json
import math
import os

LINE_LIMIT = 8192
PREFIX = "#section "

# the contract line's keys, in order; anything else in the result is a section
CONTRACT = ("metric", "value", "unit", "n_gpus", "dist_ranks", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "parity_level", "backend", "verified", "verify", "roofline",
            "pipeline_roofline", "cpu_baseline", "cpu_baseline_node", "cpu_model")


def finite(x, digits=7):
    """x with every float rounded to `digits` significant digits and every non-finite float replaced by None"""
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {str(k): finite(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [finite(v, digits) for v in x]
    return x


DEEPER = "(in the side file)"


def brief(x, digits=5, maxlen=96, depth=0, maxdepth=3):
    """the stdout copy of a section: floats to `digits` significant digits, long explanatory strings cut, objects nested deeper than
    `maxdepth` replaced by a pointer (the side file has everything whole)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if math.isfinite(x) else None
    if isinstance(x, str):
        return x if len(x) <= maxlen else x[:maxlen - 3] + "..."
    if isinstance(x, dict):
        if depth >= maxdepth:
            return DEEPER
        return {str(k): brief(v, digits, maxlen, depth + 1, maxdepth) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [brief(v, digits, maxlen, depth, maxdepth) for v in x]
    return x


def _get(d, *path):
    for p in path:
        if not isinstance(d, dict) or p not in d:
            return None
        d = d[p]
    return d


def summary(res):
    """one number per section of the default line (None where the section did not run)"""
    s = {
        "ntt_fwd_per_s": _get(res, "ntt", "steady_state", "forward", "per_s"),
        "ntt_inv_per_s": _get(res, "ntt", "steady_state", "inverse", "per_s"),
        "ntt_fwd_frac": _get(res, "ntt", "steady_state", "forward", "frac_of_hbm_peak"),
        "ntt_inv_frac": _get(res, "ntt", "steady_state", "inverse", "frac_of_hbm_peak"),
        "ntt_N": _get(res, "ntt", "steady_state", "N"),
        "c2_fwd_per_s": _get(res, "c2", "forward", "per_s"),
        "c2_inv_per_s": _get(res, "c2", "inverse", "per_s"),
        "c2_fwd_frac": _get(res, "c2", "forward", "roofline", "frac"),
        "c2_inv_frac": _get(res, "c2", "inverse", "roofline", "frac"),
        "c2_verified_limbs": _get(res, "c2", "verified_limbs"),
        "coeffwise_mul_frac": _get(res, "coeffwise", "mul", "frac_of_hbm_peak"),
        "coeffwise_add_frac": _get(res, "coeffwise", "add", "frac_of_hbm_peak"),
        "bgv_per_s": _get(res, "bgv", "per_s"),
        "bgv_A_step_frac": _get(res, "bgv", "pipeline_roofline", "frac_of_hbm_peak"),
        "level_a_per_s": _get(res, "level_a", "ckks", "per_s"),
        "level_a_roofline_frac": _get(res, "level_a", "ckks", "roofline", "frac"),
        "level_a_bgv_per_s": _get(res, "level_a", "bgv", "per_s"),
        "object_api_single_per_s": _get(res, "object_api", "single_calls", "per_s"),
        "object_api_batched_per_s": _get(res, "object_api", "batched_call", "per_s"),
        "object_api_unchanged_loop_per_s": _get(res, "object_api", "unchanged_loop", "per_s"),
        "object_api_devices": _get(res, "object_api", "devices", "summary"),
        "rotate_bench_ms_N32768": _get(res, "object_api", "reference_benchmark", "by_N", "32768", "back_to_back"),
        "step_traffic_bytes_per_op": _get(res, "step", "step_traffic", "measured_bytes_per_op"),
        "step_kernel_ms": _get(res, "step", "kernel_ms_per_step"),
        "hbm_copy_ceiling_GBps": res.get("hbm_copy_ceiling_GBps"),
        "hbm_stream_ceiling_GBps": res.get("hbm_stream_ceiling_GBps"),
        "sclk_MHz": _get(res, "chip", "timed_region", "sclk_MHz"),
        "socket_power_W": _get(res, "chip", "timed_region", "socket_power_W"),
        "all_sections_verified": res.get("all_sections_verified"),
    }
    return {k: v for k, v in s.items() if v is not None}


def _trim(entry, keep, maxlen=200):
    if not isinstance(entry, dict):
        return entry
    out = {k: entry[k] for k in keep if k in entry}
    for k, v in out.items():
        if isinstance(v, str) and len(v) > maxlen:
            out[k] = v[:maxlen - 3] + "..."
    return out


def contract_line(res, sections_file):
    """the compact last line (dict): the driver's contract fields + roofline + cpu_baseline + one-number summaries"""
    line = {k: res[k] for k in CONTRACT if k in res}
    if "verify" in line:
        line["verify"] = _trim(line["verify"], ("outputs_compared_per_gpu", "checker_evaluations", "checker"))
    if "roofline" in line and isinstance(line["roofline"], dict):
        roof = dict(line["roofline"])
        if isinstance(roof.get("traffic_source"), str) and len(roof["traffic_source"]) > 160:
            roof["traffic_source"] = roof["traffic_source"][:157] + "..."
        line["roofline"] = roof
    for k in ("cpu_baseline", "cpu_baseline_node"):
        if k in line:
            line[k] = _trim(line[k], ("value", "unit", "cores", "kind", "sample", "per_process_min", "per_process_max", "cores_visible",
                                      "cpu_quota_cores", "parallel_efficiency", "host_limited", "error"), 240)
    if "rccl" in res:
        line["rccl"] = _trim(res["rccl"], ("initialised", "backend", "init_s", "error"), 120)
    sm = summary(res)
    if sm:
        line["summary"] = sm
    names = section_names(res)
    if names:
        line["sections"] = {"names": names, "file": sections_file, "stdout_prefix": PREFIX.strip()}
    return finite(line, 10)


def section_names(res):
    return [k for k in res if k not in CONTRACT and k not in ("rccl",)]


STDOUT_LIMIT = 24576   # everything bench.py prints (sections + contract line), asserted by the tests


def emit(res, out):
    """print the sections, write the side file, print the contract line LAST; returns the line's text"""
    path = os.environ.get("HP_BENCH_SECTIONS", "bench_sections.json")
    full = finite(res, 9)
    names = section_names(res)
    wrote = None
    if path and names:
        try:
            with open(path, "w") as f:
                json.dump(full, f, allow_nan=False)
                f.write("\n")
            wrote = path
        except OSError as e:    # a read-only working directory must not void the number: the stdout lines still carry the sections
            wrote = f"not written: {e!r}"[:120]
    # stdout carries a BRIEF copy of every section (5 significant digits, explanatory strings cut): everything bench.py prints stays well
    # inside 32 KiB, so a reader that keeps only the tail of the output still has whole lines; the side file has every digit and word
    for k in names:
        print(PREFIX + k + " " + json.dumps(brief(full[k]), allow_nan=False, separators=(",", ":")), file=out)   # (brief OF the side file's values)
    line = contract_line(res, wrote)
    text = json.dumps(line, allow_nan=False)
    if len(text.encode()) >= LINE_LIMIT:     # never again a line the driver cannot read: drop the optional parts, loudly
        for k in ("sections", "summary", "cpu_baseline_node", "rccl", "verify"):
            line.pop(k, None)
            line["trimmed"] = True
            text = json.dumps(line, allow_nan=False)
            if len(text.encode()) < LINE_LIMIT:
                break
    assert len(text.encode()) < LINE_LIMIT, len(text)
    out.flush()
    print(text, file=out)
    out.flush()
    return text


def _no_constants(name):
    raise ValueError(f"non-finite constant {name} in a bench line")


def strict_loads(text):
    """json.loads that refuses NaN / Infinity"""
    return json.loads(text, parse_constant=_no_constants)


def collect(stdout):
    """(contract line as the driver sees it, full result = contract fields + sections) from a run's stdout.
    The contract line is the LAST non-empty line, parsed strictly, alone."""
    lines = [l for l in stdout.splitlines() if l.strip()]
    if not lines:
        raise ValueError("no output")
    last = lines[-1]
    if len(last.encode()) >= LINE_LIMIT:
        raise ValueError(f"contract line is {len(last.encode())} bytes (limit {LINE_LIMIT})")
    line = strict_loads(last)
    full = dict(line)
    for l in lines[:-1]:
        if l.startswith(PREFIX):
            name, _, body = l[len(PREFIX):].partition(" ")
            full[name] = strict_loads(body)
    return line, full
