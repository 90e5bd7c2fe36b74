#!/usr/bin/env python3
"""Turns a rocprofv3 --pmc pass over bench.py into the traffic record bench.py reports (profiles/score_traffic.json).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum -d gpurun_out/pmc_c2 -o pmc --output-format csv -- \
        python3 bench.py --config c2 --steps 5 --warmup 1 --no-cpu
    python3 tools/traffic_from_pmc.py c2 score_polar_kernel 100000 gpurun_out/pmc_c2 profiles/r02_pmc_c2.txt

Read bytes per launch = RDREQ_128B x 128 + (RDREQ - RDREQ_128B) x 64 (MI355X_MICROARCH.md, HBM section: the L2's
memory-side request counters; FETCH_SIZE tallies the 128-byte requests at 64 bytes on gfx950, so it is not used).  These
are the requests the L2s send to the fabric: what the Infinity Cache serves is among them.  A second pass
(SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE) gives the issue figures bench.py reports beside
the traffic, a third (TA_BUSY_avr alone) how busy the L1 address path is, a fourth (SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE) the
vector-memory wave-instructions: x 16 address-path cycles each (the floor of a 64-lane gather, tools/ta_cost.hip) over the
CU-cycles of the launch = `bound_unit_frac`, how far the L1 address path is from its own floor.  Counters are summed over the XCDs per dispatch and over all dispatches of the scoring kernels, and divided by the number
of scoring launches.  The record stores a hash of the kernel sources: bench.py ignores it once they change.
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
    # (pass, kernel, dispatch) -> counter -> value summed over the XCDs / shader engines
    per = defaultdict(lambda: defaultdict(float))
    for f in files:
        # (GRBM_GUI_ACTIVE is collected by two passes: the fourth pass's copy gets a name of its own)
        vm_pass = os.sep + "vmem" + os.sep in f
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat in row["Kernel_Name"]:
                    cn = row["Counter_Name"]
                    if vm_pass and cn == "GRBM_GUI_ACTIVE":
                        cn = "GRBM_GUI_ACTIVE_vmem_pass"
                    per[(f, row["Kernel_Name"].split("(")[0], row["Dispatch_Id"])][cn] += float(row["Counter_Value"])
    if not per:
        raise SystemExit(f"no dispatch of {pat} in {files}")
    # mean per dispatch of every kernel, then the kernels of one scoring launch added up (a mixed launch runs two)
    by_kernel = defaultdict(lambda: defaultdict(list))
    for (_, kern, _), ctr in per.items():
        for c, v in ctr.items():
            by_kernel[kern][c].append(v)
    names = sorted(by_kernel)
    # launches = the dispatches of the kernel every launch runs; a kernel that ran in only a few of them (the first call of
    # a run, before the caller's context knows the table's factors, takes another instantiation) counts with that share
    n_launches = max(max(len(v) for v in by_kernel[kern].values()) for kern in names)
    launch = defaultdict(float)
    for kern in names:
        for c, v in by_kernel[kern].items():
            launch[c] += sum(v) / n_launches
    m128, mrd = launch.get("TCC_EA0_RDREQ_128B_sum", 0.0), launch.get("TCC_EA0_RDREQ_sum", 0.0)
    bytes_per_launch = m128 * 128 + max(0.0, mrd - m128) * 64
    from bench import kernel_source_hash
    from top_down_renderer_amd import synth
    c = synth.CONFIGS[cfg]
    wave_samples = -(-n_launch // 64) * c.nb * c.nr
    issue = None
    if "SQ_ACTIVE_INST_VALU" in launch and "GRBM_GUI_ACTIVE" in launch:
        # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_ACTIVE_INST_VALU counts 4-cycle issue slots summed over the 1024 SIMDs;
        # SQ_LDS_IDX_ACTIVE counts cycles summed over the 256 LDS arrays.  The counter pass runs the kernels of a launch
        # one after the other, so the cycles are the sum of theirs.
        cycles = launch["GRBM_GUI_ACTIVE"] / 8.0
        issue = {"valu_busy": launch["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),
                 "lds_busy": (launch["SQ_LDS_IDX_ACTIVE"] / (256.0 * cycles)) if "SQ_LDS_IDX_ACTIVE" in launch else None,
                 "insts_per_sample": (launch["SQ_INSTS_VALU"] / wave_samples) if "SQ_INSTS_VALU" in launch else None,
                 # TA_BUSY_avr (a pass of its own): busy cycles of the texture addressers — the L1 address path — averaged
                 # over the CUs, against the kernels' cycles
                 "l1_addr_busy": (launch["TA_BUSY_avr"] / cycles) if "TA_BUSY_avr" in launch else None,
                 "cycles": cycles, "wave_samples": wave_samples}
        if "SQ_INSTS_VMEM_RD" in launch:
            # vector-memory READ wave-instructions (gathers) of a launch; the same pass's own cycle count where it has one
            vcyc = launch.get("GRBM_GUI_ACTIVE_vmem_pass", 0.0) / 8.0 or cycles
            issue["vmem_rd_insts"] = launch["SQ_INSTS_VMEM_RD"]
            issue["bound_unit_frac"] = launch["SQ_INSTS_VMEM_RD"] * 16.0 / (256.0 * vcyc)
            issue["vmem_per_sample"] = launch["SQ_INSTS_VMEM_RD"] / wave_samples
    path = os.path.join(ROOT, "profiles", "score_traffic.json")
    try:
        rec = json.load(open(path))
        assert "entries" in rec
    except Exception:
        rec = {"entries": {}}
    rec["entries"][cfg] = {
        "kernel": " + ".join(names), "particles_per_launch": n_launch, "hbm_bytes_per_launch": bytes_per_launch,
        "dispatches": len(per), "TCC_EA0_RDREQ_128B_sum": m128, "TCC_EA0_RDREQ_sum": mrd, "issue": issue,
        "kernel_source_hash": kernel_source_hash(), "source": os.path.relpath(summary, ROOT),
        "how": "rocprofv3 --pmc, passes of their own over bench.py (counters only); all dispatches of the scoring kernels "
               "added up and divided by the number of scoring launches; bytes = RDREQ_128B x 128 + (RDREQ - RDREQ_128B) x 64: requests "
               "the L2s send to the fabric — hits in the 256 MB Infinity Cache are among them, so this is an upper bound on "
               "what HBM itself delivers",
    }
    json.dump(rec, open(path, "w"), indent=1)
    with open(summary, "w") as fh:
        fh.write(f"{cfg}: kernels {' + '.join(names)}, {len(per)} dispatches, {n_launch} particles per launch\n")
        for kern in names:
            fh.write(f"  {kern}\n")
            for cn, v in sorted(by_kernel[kern].items()):
                fh.write(f"     {cn:32s} n={len(v)} mean={sum(v) / len(v):.6g}\n")
        fh.write(f"L2 -> fabric read bytes per launch {bytes_per_launch:.6g}  ({bytes_per_launch / n_launch:.6g} per particle; "
                 f"Infinity-Cache hits included)\n")
        if issue:
            fh.write(f"issue: vector units busy {issue['valu_busy']:.3f} (four cycles per instruction), LDS busy {issue['lds_busy']}, "
                     f"L1 address path busy {issue['l1_addr_busy']}, vector instructions per wave-sample {issue['insts_per_sample']}\n")
            if "bound_unit_frac" in issue:
                fh.write(f"vector-memory read wave-instructions per launch {issue['vmem_rd_insts']:.6g} ({issue['vmem_per_sample']:.3f} per "
                         f"wave-sample) x 16 cycles / (256 CUs x {issue['cycles']:.6g} cycles) = bound_unit_frac {issue['bound_unit_frac']:.3f}\n")
    print(open(summary).read())


if __name__ == "__main__":
    main()
