"""Build profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over one bench.py step.

    python3 tools/pmc_traffic.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json>

FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B per the guide; gfx950 reports half of the bytes of 16-B/lane
streaming reads (MI355X_MICROARCH.md, HBM section), hence the x2 on FETCH for kernels that read with dwordx4.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    path = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [set(), 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        key = name[:name.index(">") + 1] if ("gemm_nt" in name and ">" in name) else name.split("(")[0]
        agg[key][0].add(r["Dispatch_Id"])
        agg[key][1] += float(r["Counter_Value"])
    return {k: (len(v[0]), v[1]) for k, v in agg.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    allk = {}
    for k in sorted(set(fetch) | set(write)):
        allk[k] = {"launches": fetch.get(k, write.get(k))[0], "FETCH_SIZE_KB": fetch.get(k, (0, 0.0))[1],
                   "WRITE_SIZE_KB": write.get(k, (0, 0.0))[1]}
    # the dominant kernel: the 128 x 128-tile launches of the MFMA GEMM, incl. the trapezoid launch of the factorisation (same tile body)
    dom = [k for k in allk if "gemm_nt_f64_kernel<4, 4" in k or "gemm_nt_f64_trap_signal_kernel" in k or "gemm_nt_f64_reduce_kernel" in k]
    launches = sum(allk[k]["launches"] for k in dom)
    f_raw = sum(allk[k]["FETCH_SIZE_KB"] for k in dom) * 1024.0
    w = sum(allk[k]["WRITE_SIZE_KB"] for k in dom) * 1024.0
    out = {
        "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-propagate --no-extras --no-python-api",
        "workload": "C3: N=16384 d=8 M=16384; every fit+estimate_many step the command runs (timed + untimed profiling step)",
        "kernel": "gemm_nt_f64_kernel<4,4,*> + gemm_nt_f64_trap_signal_kernel + gemm_nt_f64_reduce_kernel (128x128 tile launches: the bulk of the flops)",
        "launches": launches,
        "FETCH_SIZE_bytes_raw": f_raw,
        "FETCH_SIZE_bytes_corrected": 2.0 * f_raw,
        "WRITE_SIZE_bytes": w,
        "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM); the kernel stages "
                      "with global_load_lds_dwordx4.  Calibration inside the same run: gram_kernel "
                      "(16-B stores) has a known algorithmic byte count (2 x 2.15 GB + 1.07 GB written per step), see all_kernels_raw.",
        "hbm_bytes_per_launch": (2.0 * f_raw + w) / max(launches, 1),
        "all_kernels_raw": allk,
    }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("launches", "hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main()
