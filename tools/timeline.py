"""Timeline of the LAST proof in a rocprofv3 kernel trace of bench.py: kernels in start order with their gaps.
Usage: python tools/timeline.py <bench_kernel_trace.csv> [proofs_in_trace]
A proof is delimited by its four msm_accumulate launches (the commitment groups of rounds 1, 2, 3, 5)."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
acc = [i for i, r in enumerate(rows) if "msm_accumulate" in r["Kernel_Name"]]
last4 = acc[-4:]
prev_end = acc[-5] if len(acc) >= 5 else 0
# the proof starts after the previous proof's last kernel: walk forward from the previous proof's 4th accumulate to the
# first wire kernel (largest gap before the next accumulate)
seg = rows[prev_end + 1:]
gaps = [(int(seg[k + 1]["Start_Timestamp"]) - int(seg[k]["End_Timestamp"]), k) for k in range(min(len(seg) - 1, last4[0] - prev_end - 1))]
start = max(gaps)[1] + 1 if gaps else 0
seg = seg[start:]
t0 = int(seg[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in seg)
print(f"last proof: {len(seg)} kernels, {(end - t0) / 1e6:.3f} ms from first kernel start to last kernel end")
busy = 0
cur_end = t0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - cur_end
    name = r["Kernel_Name"].split("(")[0].replace("plonk::", "")[:44]
    flag = f"   <-- idle {gap / 1e3:7.1f} us" if gap > 15000 else ""
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {name}{flag}")
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
print(f"device busy {busy / 1e6:.3f} ms of {(end - t0) / 1e6:.3f} ms")
