"""Per-launch averages of the rocprofv3 PMC passes over tools/pmc_target.py -> summary.json.

usage: python tools/pmc_summarize.py <dir with *counter_collection.csv> <out summary.json>

Counters are summed over XCDs/instances by rocprofv3 already (one row per dispatch and counter).
HBM bytes: the CSV holds the derived FETCH_SIZE / WRITE_SIZE metrics in KB, so they are multiplied by
1024 here.  WRITE_SIZE is exact for 16-B/lane streaming stores.  On gfx950 FETCH_SIZE counts half of a wide coalesced streaming read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so both the raw value ("lower") and the
doubled value ("upper") are kept; bench.py picks per kernel.
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

KERNELS = {"k_db_scan": r"k_db_scan<(8, )?false(, 4)?>", "k_hamming_matrix": r"k_hamming_matrix\(",
           "k_db_scan_rows_q1": r"k_db_scan_rows<4, ", "k_db_scan_rows_q8": r"k_db_scan_rows<8, ", "k_db_scan_rows_q32": r"k_db_scan_rows<16, "}
NAMES = {"FETCH_SIZE": "fetch_size_raw_bytes", "WRITE_SIZE": "write_size_bytes", "SQ_INSTS_VALU": "valu_wave_insts",
         "GRBM_GUI_ACTIVE": "grbm_gui_active_sum", "SQ_WAVE_CYCLES": "sq_wave_cycles", "SQ_WAIT_ANY": "sq_wait_any",
         "SQ_WAIT_INST_ANY": "sq_wait_inst_any", "SQ_ACTIVE_INST_ANY": "sq_active_inst_any", "SQ_WAVES": "waves",
         "SQ_INSTS_LDS": "lds_insts", "SQ_LDS_BANK_CONFLICT": "lds_bank_conflict", "SQ_INSTS_SMEM": "smem_insts"}


def main(src, dst):
    acc = {k: defaultdict(list) for k in KERNELS}
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            for k, pat in KERNELS.items():
                if re.search(pat, row["Kernel_Name"]):
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, ctr in acc.items():
        d = {}
        for cname, vals in ctr.items():
            v = sum(vals) / len(vals)
            if cname in ("FETCH_SIZE", "WRITE_SIZE"):
                v *= 1024.0                       # derived metric is in KB
            d[NAMES.get(cname, cname)] = v
        if "fetch_size_raw_bytes" in d and "write_size_bytes" in d:
            d["hbm_bytes_lower"] = d["fetch_size_raw_bytes"] + d["write_size_bytes"]
            d["hbm_bytes_upper"] = 2 * d["fetch_size_raw_bytes"] + d["write_size_bytes"]
        if "grbm_gui_active_sum" in d:
            d["cycles_per_xcd"] = d["grbm_gui_active_sum"] / 8
        out[k] = d
    out["_how"] = ("rocprofv3 --pmc <one group per pass> --kernel-trace --output-format csv -- python3 tools/pmc_target.py; "
                   "tools/pmc_summarize.py; shapes: k_db_scan L=10000 n=64 Q=500 (algorithmic 20,536,000 B), "
                   "k_hamming_matrix 20000x20000 (algorithmic 801,280,000 B), k_db_scan_rows L=100000 n=64 Q=1 / 8 / 32 (algorithmic 205,200,032 / 205,200,256 / 205,201,024 B)")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
