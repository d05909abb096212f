"""Turn a gpurun_out/prof_<tag> directory (tools/profile_bench.sh) into profiles/<tag>/ (tracked)."""
import collections
import csv
import os
import shutil
import sys

tag = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else ""
base = f"gpurun_out/prof_{tag}/"
dst = f"profiles/{tag}/"
os.makedirs(dst, exist_ok=True)
shutil.copy(base + "trace/bench_kernel_stats.csv", dst + "bench_kernel_stats.csv")
out = [f"# {tag} — rocprofv3 summary of `python bench.py` (2^20 gates, 1x MI355X)\n"]
if note:
    out.append(note + "\n")
out.append("Commands (on the GPU box, tools/profile_bench.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --steps 5 --warmup 2`; "
           "PMC in separate passes `rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` / `--pmc WRITE_SIZE ...` / SQ counters (1 proof each).")
out.append("Counts include the one-off setup (SRS generation + window tables, key commitments, 16 key coset NTTs) and the warm-up + timed proofs (7 in total).\n")
out.append("## Kernel time (--kernel-trace --stats)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in list(csv.DictReader(open(base + "trace/bench_kernel_stats.csv")))[:22]:
    out.append("| `%s` | %s | %.3f | %.1f | %s |" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                float(r["AverageNs"]) / 1e3, r["Percentage"]))
try:   # the launches that belong to proofs (bench.py's roofline leg times exactly these): 4 per proof, at the end
    tr = sorted(csv.DictReader(open(base + "trace/bench_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
    acc_d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr if "msm_accumulate" in r["Kernel_Name"]]   # any variant (ordered lanes by default)
    nproofs = (len(acc_d) - 4) // 4          # setup commits the 15 key polynomials in 4 group launches
    prove_d = acc_d[-4 * nproofs:]
    names = collections.Counter(r["Kernel_Name"].split("(")[0] for r in tr if "msm_accumulate" in r["Kernel_Name"])
    out.append(f"\n`{names.most_common(1)[0][0]}` (every msm_accumulate launch of this run: {dict(names)}) — launches inside prove() only ({len(prove_d)} launches = {nproofs} proofs x 4 commitment groups): "
               f"average **{sum(prove_d) / len(prove_d) / 1e6:.3f} ms** per launch, {4 * sum(prove_d) / len(prove_d) / 1e6:.2f} ms per proof "
               "(compare `roofline.avg_launch_ms` / `kernel_ms_per_prove.msm_accumulate` printed by bench.py).")
except Exception as e:  # noqa
    out.append(f"\n(per-proof accumulate average unavailable: {e})")
pm = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(base + f"pmc_{C}/bench_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0][:70]              # the kernel that RAN, namespace included (nb15:: / nbl::, ordered lanes or not)
        agg[name].append(float(r["Counter_Value"]))
    pm[C] = agg
    out.append(f"\n## {C} per launch (raw counter value, KiB)\n\n| kernel | launches | avg per launch |\n|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:10]:
        out.append("| `%s` | %d | %.1f |" % (k, len(v), sum(v) / len(v)))
try:
    rows = list(csv.DictReader(open(base + "pmc_SQ/bench_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    out.append("\n## SQ counters (summed over launches of 1 proof + setup)\n\n| kernel | SQ_WAVES | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU | SQ_WAIT_INST_ANY | SQ_WAVE_CYCLES | SQ_BUSY_CYCLES |\n|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
        out.append("| `%s` | %s |" % (k, " | ".join("%.3g" % v.get(c, 0) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"))))
except Exception as e:  # noqa
    out.append(f"\n(SQ counters unavailable: {e})")
# the dominant kernel under the name it ran as: the msm_accumulate variant with the most launches in the counter pass
cands = [k for k in pm["FETCH_SIZE"] if "msm_accumulate" in k]
acc = max(cands, key=lambda k: len(pm["FETCH_SIZE"][k])) if cands else ""
if acc in pm["FETCH_SIZE"]:
    f = pm["FETCH_SIZE"][acc]; w = pm["WRITE_SIZE"].get(acc, [0])
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    out.append(f"\n## HBM traffic of the dominant kernel (`{acc}`, per launch = one commitment group)\n")
    out.append(f"* raw FETCH_SIZE {fa:,.0f} KiB, WRITE_SIZE {wa:,.0f} KiB per launch (average over groups of 4/1/4/2 MSMs).")
    out.append(f"* MI355X_MICROARCH.md §HBM correction (gfx950 FETCH_SIZE = 1/2 of wide-read bytes): read = 2 x FETCH = {2 * fa * 1024 / 1e9:.2f} GB, "
               f"+ written {wa * 1024 / 1e9:.2f} GB = **{(2 * fa + wa) * 1024 / 1e9:.2f} GB per launch** (upper bound — the gather pattern here is 4 x 16 B per lane "
               f"from random 128-B table entries, for which the guide gives no calibration; uncorrected it is {(fa + wa) * 1024 / 1e9:.2f} GB).")
    out.append("* algorithmic bytes per launch (bench.py): (32 b + 96) m averaged over the groups = 0.193 GB.  The excess is by design: one precomputed "
               "table row per digit position is gathered (~12.1 x 128 B per term with bit-position tables and 2^19 buckets, 14.7 with 2^15 buckets, 16 x 128 B with window tables) so that all digits "
               "share one bucket set; the kernel is integer-VALU bound, not HBM bound.")
    import json
    valu = None
    try:
        a = agg[acc]
        # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES both count quad-cycles summed over waves (MI355X_MICROARCH.md,
        # "s_memtime tick vs SQ PMC units"); the kernel runs 2 waves per SIMD (216 VGPRs), so SIMD time =
        # WAVE_CYCLES / 2 and VALU-busy = ACTIVE_INST_VALU / (WAVE_CYCLES / 2).
        valu_raw = a["SQ_ACTIVE_INST_VALU"] / (a["SQ_WAVE_CYCLES"] / 2)
        valu = min(valu_raw, 1.0)   # waves that retire early make the 2-waves-per-SIMD denominator a slight underestimate
        out.append(f"\n## VALU utilisation of `{acc}`\n\nSQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / 2 waves per SIMD) = **{valu_raw:.2f}** — "
                   "the integer VALU pipe is saturated; only fewer instructions per point addition make this kernel faster.")
        nk = max((k for k in agg if "ntt_pass_kernel" in k), key=lambda k: agg[k]["SQ_WAVE_CYCLES"])
        b = agg[nk]
        out.append(f"Same ratio for `{nk}` (2 workgroups x 4 waves per CU = 2 waves per SIMD): {b['SQ_ACTIVE_INST_VALU'] / (b['SQ_WAVE_CYCLES'] / 2):.2f}.")
    except Exception as e:  # noqa
        out.append(f"\n(VALU utilisation unavailable: {e})")
    json.dump({"kernel": acc, "workload": "bench.py 2^20 gates, 1 GPU", "fetch_size_kib_per_launch": fa,
               "valu_busy_frac": valu, "valu_busy_formula": "SQ_ACTIVE_INST_VALU / (SQ_WAVE_CYCLES / 2 waves per SIMD), quad-cycle units",
               "write_size_kib_per_launch": wa, "traffic_bytes_per_launch": (2 * fa + wa) * 1024,
               "correction": "2 x FETCH_SIZE (gfx950, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes"},
              open(dst + "pmc.json", "w"), indent=1)
open(dst + "SUMMARY.md", "w").write("\n".join(out) + "\n")
print("\n".join(out)[:3000])
