#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output) -> txt + json.

    python tools/pmc_traffic.py <dir with *counter_collection.csv> profiles/round1_pmc_hbm_traffic

Counter values are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of
a wide coalesced 16 B/lane stream, other access widths in full.  `fetch_x2` doubles the whole counter (the guide's
prescription, an upper bound when a kernel also has narrow loads); `fetch_calibrated` doubles only the 16 B/lane part
(counter - narrow bytes) where the narrow bytes are known exactly from the kernel (scalar tap loads).
"""
import collections
import csv
import glob
import json
import os
import sys

B, C = 64, 256
PL = B * C
# kernel-name substring -> (algorithmic read bytes, algorithmic write bytes, narrow (non 16 B/lane) read bytes) per launch
KERNELS = {
    "xcorr_north_fft2_kernel": (PL * (61 * 61 + 31 * 31) * 4, PL * 31 * 31 * 4, 0),
    # the column-first kernel loads 4 B per lane, but as coalesced 244-byte rows: FETCH_SIZE comes out at 0.502 of its
    # algorithmic read bytes, i.e. the same half-counting as a 16 B/lane stream (calibrated on this kernel's own byte count)
    "xcorr_north_fft4_kernel": (PL * (61 * 61 + 31 * 31) * 4, PL * 31 * 31 * 4, 0),
    "xcorr_north_kernel": (PL * (61 * 61 + 31 * 31) * 4, PL * 31 * 31 * 4, PL * 31 * 31 * 4),   # taps via s_load
    "xcorr_north_mfma_kernel": (PL * (61 * 61 + 31 * 31) * 4, PL * 31 * 31 * 4, 0),
    "xcorr_prod29_kernel": (6 * PL * (29 * 29 + 25) * 4, 6 * PL * 25 * 25 * 4, 6 * PL * 25 * 4),
    "xcorr_circ13_kernel": (6 * PL * 2 * 169 * 4, 6 * PL * 169 * 4, 0),
    "xcorr_circ13r_kernel": (6 * PL * 2 * 169 * 4, 6 * PL * 169 * 4, 0),
    "xcorr_circ13f_kernel": (6 * PL * 2 * 169 * 4, 6 * PL * 169 * 4, 0),  # 16 B/lane words (+ one float per plane)
}


SOURCES = {"xcorr_north_fft": "xcorr_fft.hip"}     # kernel-name substring -> source file under hdn_amd/csrc (default: xcorr.hip)


def source_hash(kernel_name):
    import hashlib
    fn = next((v for k, v in SOURCES.items() if k in kernel_name), "xcorr.hip")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hdn_amd", "csrc", fn)
    return fn, (hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None)


def main(src, dst):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for name, ctr in sorted(acc.items()):
        if "FETCH_SIZE" not in ctr or "WRITE_SIZE" not in ctr:
            continue
        key = next((k for k in sorted(KERNELS, key=len, reverse=True) if k in name), None)
        fetch = 1024.0 * sum(ctr["FETCH_SIZE"]) / len(ctr["FETCH_SIZE"])
        write = 1024.0 * sum(ctr["WRITE_SIZE"]) / len(ctr["WRITE_SIZE"])
        rec = {"launches": len(ctr["FETCH_SIZE"]), "fetch_counter_bytes": fetch, "write_counter_bytes": write,
               "fetch_x2_bytes": 2 * fetch}
        if key:
            rd, wr, narrow = KERNELS[key]
            cal = 2 * (fetch - narrow) + narrow
            rec.update({"algorithmic_read_bytes": rd, "algorithmic_write_bytes": wr, "algorithmic_bytes": rd + wr,
                        "kernel_source": source_hash(name)[0], "kernel_source_sha256": source_hash(name)[1],   # bench.py checks it against the tree it runs from
                        "fetch_calibrated_bytes": cal, "traffic_calibrated_bytes": cal + write,
                        "traffic_over_algorithmic": (cal + write) / (rd + wr),
                        "how": "profiles/%s.txt: 2*(FETCH_SIZE - narrow-load bytes) + narrow-load bytes + WRITE_SIZE, per launch "
                               "(narrow = %d B of scalar tap loads); x2 on the whole FETCH_SIZE would give %.0f"
                               % (os.path.basename(dst), narrow, 2 * fetch + write)})
        out[name] = rec
    with open(dst + ".json", "w") as f:
        json.dump(out, f, indent=1)
    with open(dst + ".txt", "w") as f:
        f.write(__doc__.split("\n\n", 2)[2].replace("\n", "\n# ").join(["# ", "\n"]) if False else "")
        f.write("# HBM traffic per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, --kernel-trace only)\n"
                "# command: rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown\n"
                "# counter values are KiB; gfx950: FETCH_SIZE counts half of a 16 B/lane stream (x2), narrow loads in full; see tools/pmc_traffic.py\n\n")
        for name, rec in out.items():
            f.write(name + "\n")
            for k, v in rec.items():
                if k != "how":
                    f.write("    %-28s %s\n" % (k, ("%.4f" % v) if isinstance(v, float) and v < 100 else ("{:,.0f}".format(v) if isinstance(v, float) else v)))
    print(open(dst + ".txt").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
