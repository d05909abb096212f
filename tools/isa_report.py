"""Static ISA report of the gfx950 kernels (no GPU needed): for every kernel of the given sources the instruction mix the
issue-bound analysis of DESIGN.md rests on — VALU instructions (static count of the unrolled code), v_mad_u64_u32 among them,
LDS / global memory instructions, VGPRs, scratch bytes, LDS bytes.  `python tools/isa_report.py [out.md] [sources...]`
compiles with the flags of __graft_entry__.py (`--cuda-device-only -S`); the MSM sources are reported for both bucket counts."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return p.stdout.strip().splitlines() if p.returncode == 0 else names
    except OSError:
        return names


def report(src, extra, tag):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([G.HIPCC, *G.HIP_FLAGS, *extra, "--cuda-device-only", "-S", os.path.join(G.CSRC, src), "-o", out],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    kernels = collections.OrderedDict()
    cur = None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = collections.Counter()
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        m = re.match(r"^\s+([a-z]\w+)\s", line)
        if m and cur:
            kernels[cur][m.group(1)] += 1
    meta = {}
    for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in
                      ("vgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "sgpr_count")}
    rows = []
    names = [k for k in kernels if k in meta]
    for k, pretty in zip(names, demangle(names)):
        c = kernels[k]
        valu = sum(n for i, n in c.items() if i.startswith("v_"))
        rows.append((re.sub(r"\(.*", "", pretty).replace("void ", "").replace("plonk::", ""), valu, c["v_mad_u64_u32"],
                     sum(n for i, n in c.items() if i.startswith("ds_")), sum(n for i, n in c.items() if i.startswith("global_")),
                     meta[k]["vgpr_count"], meta[k]["private_segment_fixed_size"], meta[k]["group_segment_fixed_size"]))
    rows.sort(key=lambda r: -r[1])
    lines = [f"\n### {src} {tag}\n", "| kernel | VALU instr (static) | v_mad_u64_u32 | LDS instr | global instr | VGPRs | scratch B | static LDS B |", "|---|---|---|---|---|---|---|---|"]
    lines += ["| `%s` | %d | %d | %d | %d | %d | %d | %d |" % r for r in rows if r[1] >= 200]
    return lines


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "isa_report.md")
    srcs = sys.argv[2:] or ["msm.hip", "msm_sort.hip", "ntt.hip", "poly.hip", "prover.hip", "serial.hip"]
    lines = ["# Static ISA report (hipcc --offload-arch=gfx950 -O3, device code only)\n",
             "Counts are of the emitted code (every branch, loops counted once): they bound a kernel's instruction footprint and show its mix; "
             "the executed counts per wave come from the SQ counters in the round's SUMMARY.md files.  Kernels under 200 VALU instructions are omitted."]
    for s in srcs:
        lines += report(s, [], "(2^15 buckets)" if s.startswith("msm") else "")
        if s.startswith("msm"):
            lines += report(s, ["-DPLONK_MSM_NB_BITS=" + G.MSM_LARGE_BITS], f"(2^{G.MSM_LARGE_BITS} buckets, namespace nbl)")
    open(out, "w").write("\n".join(lines) + "\n")
    print(out)


if __name__ == "__main__":
    main()
