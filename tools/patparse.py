"""patparse.py (CPU) — splits the kernel trace of `rocprofv3 --kernel-trace --output-format csv -d gpurun_out/patprof -- python3 tools/patbench.py 100` into the variants
patbench.py runs one after the other (110 launches each): median / min / max kernel duration per variant."""
import csv, glob, statistics
f = glob.glob('gpurun_out/patprof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'][:34], int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows if 'k_pattern' in r['Kernel_Name']]
# the optimizer's own 3 steps alternate fwd/bwd; then runs of 110 identical launches per variant
i = 0
while i < len(seq):
    j = i
    while j < len(seq) and seq[j][0] == seq[i][0]: j += 1
    if j - i >= 50:
        d = [x[1] for x in seq[i + 10:j]]
        print(f"{seq[i][0]:36s} n={j-i:4d} median {statistics.median(d)/1e3:7.2f} us  min {min(d)/1e3:7.2f}")
    i = j
print()
names = ["step one launch", "step one launch slots", "bwd full", "no dot", "no reg", "no adam no dot", "dot no update", "no data", "reg only no adam", "fwd_blur", "fwd_blur no zero"]
body = seq[6:]  # the optimiser's own three steps (fwd, bwd each)
for k, nm in enumerate(names):
    d = [x[1] for x in body[k * 110 + 10:(k + 1) * 110]]
    kn = body[k * 110 + 10][0]
    print(f"{nm:20s} {kn:34s} median {statistics.median(d)/1e3:7.2f} us  min {min(d)/1e3:7.2f}  max {max(d)/1e3:7.2f}")
