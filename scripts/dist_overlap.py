#!/usr/bin/env python3
"""How much of the distributed FFT's exchange time runs under its local passes?  Reads a `rocprofv3 --kernel-trace`
CSV of `bench.py --workload fft_dist --dist-exchange rccl` and, over the last transforms of the run, reports the time
RCCL kernels were executing, the time local kernels were executing and the time both were (interval intersection).
    python scripts/dist_overlap.py <dir with *_kernel_trace.csv> [transforms [exchanges per transform]]"""
import csv
import glob
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    root = sys.argv[1]
    files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    assert files, "no kernel trace under " + root
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    comm = [(a, b) for a, b, n in rows if "nccl" in n.lower() or "rccl" in n.lower()]
    # the transforms' own kernels (everything of namespace dspgpu)
    local = [(a, b) for a, b, n in rows if "dspgpu" in n]
    if not comm:
        print("no RCCL kernels in the trace")
        return
    # keep the tail of the run: the last `transforms` transforms' worth of exchange kernels
    transforms = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    t0 = comm[-per * transforms][0]
    local_t = [iv for iv in local if iv[1] > t0]
    t0 = min(t0, min(a for a, _ in local_t))
    comm_u, local_u = union([iv for iv in comm if iv[0] >= t0]), union([iv for iv in local if iv[0] >= t0])
    both = intersect(comm_u, local_u)
    span = max(comm_u[-1][1], local_u[-1][1]) - t0
    print(f"exchange kernels per transform: {per}; window: last {transforms} transforms, {span / 1e6 / transforms:.3f} ms each")
    print(f"  exchange kernels executing : {length(comm_u) / 1e6 / transforms:.3f} ms per transform")
    print(f"  local kernels executing    : {length(local_u) / 1e6 / transforms:.3f} ms per transform")
    print(f"  both at once               : {length(both) / 1e6 / transforms:.3f} ms per transform"
          f" = {length(both) / max(length(comm_u), 1):.2f} of the exchange time")


if __name__ == "__main__":
    main()
