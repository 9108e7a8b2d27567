#!/usr/bin/env python3
"""Static instruction mix of the blend backward's hot path, from the compiler's own assembly listing:

    python tools/isa_mix.py [profiles/isa_mix.json]

Compiles monogs_amd/csrc/blend.hip to gfx950 assembly (hipcc -S, the flags of the Makefile; no GPU needed) and counts, for
``blend_backward_s_kernel<false>`` (and ``<true>``, the pose-only variant; ``blend_backward_t_kernel<false>``, round 3's, for comparison):

  * per SURVIVOR: the body of the inner (depth-2) loop without the batch flush -- fetch, alpha, per-pixel gradient factors,
    the two LDS stores;
  * per BATCH of four survivors: the flush -- LDS reads, the ten sums over the lane's four pixels, the packed DPP butterfly,
    the atomic.

Instructions are classed by what they cost to issue (tools/ubench/valu_rate.hip): plain VALU, transcendental, DPP,
v_cndmask, lane read / write, permlane swap, and the scalar / memory classes.  bench.py multiplies the counts with the
measured per-instruction issue times (profiles/valu_costs.json) into nanoseconds per survivor and compares that with the
kernel's measured time.  Stamped with the hash of the kernel sources, like the PMC files.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash() -> str:
    h = hashlib.sha256()
    for d in ("monogs_amd/csrc", "include"):
        for name in sorted(os.listdir(os.path.join(ROOT, d))):
            if name.endswith((".hip", ".h", "Makefile")):
                h.update(name.encode())
                h.update(open(os.path.join(ROOT, d, name), "rb").read())
    return h.hexdigest()[:16]


def classify(op: str, line: str) -> str:
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait_nop"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
        return "valu_trans"
    if "_dpp" in op or " row_" in line or "quad_perm" in line:
        return "valu_dpp"
    if op.startswith("v_cndmask"):
        return "valu_cndmask"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "valu_lane"
    if op.startswith("v_permlane"):
        return "valu_permlane_swap"
    if op.startswith("v_"):
        return "valu_plain"
    return "other"


def instructions(lines):
    out = []
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        out.append((t.split()[0], t))
    return out


def mix(ins):
    c = {}
    for op, line in ins:
        k = classify(op, line)
        c[k] = c.get(k, 0) + 1
    c["valu_total"] = sum(v for k, v in c.items() if k.startswith("valu_"))
    return c


def analyse(text: str, mangled_prefix: str):
    m = re.search(r"^(%s[^:\n]*):" % re.escape(mangled_prefix), text, re.M)
    if not m:
        raise SystemExit(f"kernel {mangled_prefix} not found in the listing")
    body = text[m.end():]
    body = body[:body.index("s_endpgm")]
    lines = body.split("\n")
    # the inner loop: from the 'Inner Loop Header: Depth=2' marker to the last line that is still 'in Loop: Header=<that label>'
    hdr = next(i for i, ln in enumerate(lines) if "Inner Loop Header: Depth=2" in ln)
    label = re.match(r"\s*(\.LBB\d+_\d+):", lines[hdr - 1] if lines[hdr - 1].strip().endswith(":") or ":" in lines[hdr - 1] else lines[hdr]).group(1)
    name = label[2:]                   # '.LBB7_82' -> 'BB7_82', as the 'in Loop: Header=' comments spell it
    in_loop = [i for i, ln in enumerate(lines) if f"Header={name} " in ln or f"Header={name}\t" in ln or ln.rstrip().endswith(f"Header={name}") or f"Header={name} Depth" in ln]
    lo, hi = min(in_loop + [hdr]), max(in_loop + [hdr])
    # extend to the end of the last block of the loop
    end = hi + 1
    while end < len(lines) and not re.match(r"\s*\.LBB\d+_\d+:", lines[end]):
        end += 1
    loop = lines[lo:end]
    # The loop holds `copies` survivors (round 5: unrolled by two, one register set each) and as many inlined flushes.  A flush:
    # from the start of the basic block that holds the first LDS READ after the previous flush to its atomic.
    atomics = [i for i, ln in enumerate(loop) if "global_atomic_add_f32" in ln]
    flushes, prev = [], 0
    for at in atomics:
        first_read = next(i for i in range(prev, at) if re.match(r"\s*ds_read", loop[i]))
        fs = first_read
        while fs > prev and not re.match(r"\s*(\.LBB\d+_\d+:|; %bb\.)", loop[fs]):
            fs -= 1
        flushes.append((fs, at + 1))
        prev = at + 1
    copies = len(flushes)
    keep = [True] * len(loop)
    for a, b in flushes:
        for i in range(a, b):
            keep[i] = False
    survivor = mix(instructions([ln for ln, k in zip(loop, keep) if k]))
    flush = mix(instructions(loop[flushes[0][0]:flushes[0][1]]))
    survivor = {k: (v / copies if copies > 1 else v) for k, v in survivor.items()}
    return survivor, flush, copies


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "isa_mix.json")
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "blend.s")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-DNDEBUG",
               "-fno-slp-vectorize", "-Wno-inline-asm", "-S", "--cuda-device-only",
               os.path.join(ROOT, "monogs_amd", "csrc", "blend.hip"), "-o", asm]
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    res = {"csrc_sha256": csrc_hash(), "batch_size": 4,
           "source": "hipcc -S of monogs_amd/csrc/blend.hip, inner loop of blend_backward_s_kernel, and of its round-3 A/B partner blend_backward_t_kernel (tools/isa_mix.py)"}
    for tag, prefix in (("blend_backward_s_kernel<false>", "_ZN3mgs23blend_backward_s_kernelILb0EE"),
                        ("blend_backward_s_kernel<true>", "_ZN3mgs23blend_backward_s_kernelILb1EE"),
                        ("blend_backward_t_kernel<false>", "_ZN3mgs23blend_backward_t_kernelILb0EE")):
        surv, flush, copies = analyse(text, prefix)
        res[tag] = {"per_survivor": surv, "per_batch": flush, "survivors_per_loop_trip": copies}
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
