#!/usr/bin/env python3
"""rocprofv3 --pmc counter CSVs of tools/pmc_gemm.py -> one table (stdout): per GEMM shape and operand mode, MFMA-busy
share of the kernel, vector instructions per MFMA, parked / issue-stalled / issuing share of wave time, LDS activity.

  python tools/pmc_gemm_summary.py gpurun_out/<dir>/passA_counter_collection.csv [passB_counter_collection.csv ...]"""
import csv
import re
import sys
from collections import defaultdict

SHAPES = {  # (mode-independent) grid size -> label; 64x64 tiles unless the 128x128 configuration is picked
}
MODES = {"0": "split on use", "1": "pre-split", "2": "single bf16"}


def main():
    acc = defaultdict(lambda: defaultdict(list))      # (kernel key) -> counter -> values
    for path in sys.argv[1:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                m = re.search(r"gemm_bf16x3_kernel<(\d+), (\d+), (\w+), (\w+), (\d)>", name)
                if not m:
                    continue
                key = (m.group(1), m.group(3), m.group(4), m.group(5), r["Grid_Size"])
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for key, c in acc.items():
        tile, ta, tb, mode, grid = key
        mean = {k: sum(v) / len(v) for k, v in c.items()}
        cyc = mean.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                     # summed over the 8 XCDs
        mfma_busy = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)            # SIMD-cycles with the matrix pipe busy
        nmfma = mean.get("SQ_INSTS_MFMA", 0.0)
        wave = mean.get("SQ_WAVE_CYCLES", 0.0)
        rows.append((grid, tile, ta, tb, mode, cyc,
                     100.0 * mfma_busy / (cyc * 1024.0) if cyc else float("nan"),
                     mean.get("SQ_INSTS_VALU", 0.0) / nmfma if nmfma else float("nan"),
                     100.0 * mean.get("SQ_WAIT_ANY", 0.0) / wave if wave else float("nan"),
                     100.0 * mean.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else float("nan"),
                     100.0 * mean.get("SQ_ACTIVE_INST_ANY", 0.0) / wave if wave else float("nan"),
                     mean.get("SQ_INSTS_LDS", 0.0) / nmfma if nmfma else float("nan"),
                     100.0 * mean.get("SQ_LDS_BANK_CONFLICT", 0.0) / mean["SQ_LDS_IDX_ACTIVE"] if mean.get("SQ_LDS_IDX_ACTIVE") else float("nan")))
    rows.sort(key=lambda r: (int(r[0]), r[4]))
    print(f"{'grid':>8} {'tile':>4} {'TA':>5} {'TB':>5} {'mode':>13} {'kcycles':>8} {'MFMA busy %':>11} {'VALU/MFMA':>9} {'parked %':>8} "
          f"{'stalled %':>9} {'issuing %':>9} {'LDS/MFMA':>8} {'LDS confl %':>11}")
    for r in rows:
        print(f"{r[0]:>8} {r[1]:>4} {r[2]:>5} {r[3]:>5} {MODES.get(r[4], r[4]):>13} {r[5] / 1e3:8.1f} {r[6]:11.1f} {r[7]:9.1f} {r[8]:8.1f} "
              f"{r[9]:9.1f} {r[10]:9.1f} {r[11]:8.2f} {r[12]:11.1f}")
    print("\\nMFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); parked = SQ_WAIT_ANY, stalled = SQ_WAIT_INST_ANY,\\n"
          "issuing = SQ_ACTIVE_INST_ANY, all over SQ_WAVE_CYCLES; VALU/MFMA counts SQ_INSTS_VALU (which includes the MFMAs) per SQ_INSTS_MFMA.")


if __name__ == "__main__":
    main()
