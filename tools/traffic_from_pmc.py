#!/usr/bin/env python3
"""Turns a rocprofv3 --pmc pass over bench.py into the traffic record bench.py reports (profiles/score_traffic.json).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum -d gpurun_out/pmc_c2 -o pmc --output-format csv -- \
        python3 bench.py --config c2 --steps 5 --warmup 1 --no-cpu
    python3 tools/traffic_from_pmc.py c2 score_polar_kernel 100000 gpurun_out/pmc_c2 profiles/r02_pmc_c2.txt

HBM read bytes per launch = RDREQ_128B x 128 + (RDREQ - RDREQ_128B) x 64 (MI355X_MICROARCH.md, HBM section: the L2's
memory-side request counters; FETCH_SIZE tallies the 128-byte requests at 64 bytes on gfx950, so it is not used).
Counters are summed over the XCDs per dispatch and averaged over the dispatches of the kernel.  The record stores a hash
of the kernel sources: bench.py ignores it once they change.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cfg, pat, n_launch, d, summary = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = defaultdict(lambda: defaultdict(float))
    names = set()
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat in row["Kernel_Name"]:
                    per[(f, row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
                    names.add(row["Kernel_Name"])
    if not per:
        raise SystemExit(f"no dispatch of {pat} in {files}")
    rd128 = [v.get("TCC_EA0_RDREQ_128B_sum", 0.0) for v in per.values()]
    rd = [v.get("TCC_EA0_RDREQ_sum", 0.0) for v in per.values()]
    m128, mrd = sum(rd128) / len(rd128), sum(rd) / len(rd)
    bytes_per_launch = m128 * 128 + max(0.0, mrd - m128) * 64
    from bench import kernel_source_hash
    path = os.path.join(ROOT, "profiles", "score_traffic.json")
    try:
        rec = json.load(open(path))
        assert "entries" in rec
    except Exception:
        rec = {"entries": {}}
    kname = sorted(names)[0].split("(")[0]
    rec["entries"][cfg] = {
        "kernel": kname, "particles_per_launch": n_launch, "hbm_bytes_per_launch": bytes_per_launch,
        "dispatches": len(per), "TCC_EA0_RDREQ_128B_sum": m128, "TCC_EA0_RDREQ_sum": mrd,
        "kernel_source_hash": kernel_source_hash(), "source": os.path.relpath(summary, ROOT),
        "how": "rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum, own pass over bench.py; "
               "bytes = RDREQ_128B x 128 + (RDREQ - RDREQ_128B) x 64, mean over the kernel's dispatches",
    }
    json.dump(rec, open(path, "w"), indent=1)
    with open(summary, "w") as fh:
        fh.write(f"{cfg}: kernel {kname}, {len(per)} dispatches, {n_launch} particles per launch\n"
                 f"TCC_EA0_RDREQ_128B_sum mean {m128:.6g}\nTCC_EA0_RDREQ_sum      mean {mrd:.6g}\n"
                 f"HBM read bytes per launch {bytes_per_launch:.6g}  ({bytes_per_launch / n_launch:.6g} per particle)\n")
    print(open(summary).read())


if __name__ == "__main__":
    main()
