"""Per-kernel MFMA utilisation / stall summary from a rocprofv3 --pmc pass (SQ_* + GRBM_GUI_ACTIVE).

    python tools/pmc_util.py gpurun_out/pmc_util profiles/r01_pmc_mfma_util.json

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs) -- rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8
XCDs (checked: value/8 = kernel duration x clock; MFMA busy = 16 cycles x SQ_INSTS_MFMA for 16x16x32 bf16); SQ_WAVE_CYCLES / SQ_WAIT_* are in
quad-cycles summed over waves (MI355X_MICROARCH.md, rocprofv3 PMC slots), reported as fractions of SQ_WAVE_CYCLES."""
import collections
import csv
import json
import sys


def short(name):
    for key, tag in (("gemm256_kernel<0, float", "gemm256 plain f32 out (logits)"), ("gemm256_kernel<0", "gemm256 plain bf16 out (FFN w3 residual producer)"), ("gemm256_kernel<1", "gemm256 SwiGLU"),
                     ("gemm256_kernel<2", "gemm256 head-split QKV"), ("gemm2b_kernel", "gemm2b (short-K residual)"), ("gemm_nt_kernel", "gemm128 (small)"),
                     ("attention_bf16_kernel", "attention"), ("attention_kernel", "attention (f32 / round-2 kernel)"), ("ln_coef_parts_kernel", "ln_coef_parts (fold coefficients from producer statistics)"), ("ln_coef_kernel", "ln_coef (fold coefficients)"),
                     ("hilo_rows_kernel", "hi/lo row operators"), ("layernorm", "layernorm"), ("sample_rows", "sample_rows"), ("vq_scan", "vq_scan")):
        if key in name:
            return tag
    return None


def main(d, out):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f"{d}/p_counter_collection.csv")):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    res = {}
    for k, v in agg.items():
        wc = v["SQ_WAVE_CYCLES"] or 1.0
        res[k] = {
            "dispatches": len(cnt[k]),
            "mfma_util": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024) if v["GRBM_GUI_ACTIVE"] else 0.0, 4),
            "wait_any_frac": round(v["SQ_WAIT_ANY"] / wc, 3), "wait_inst_any_frac": round(v["SQ_WAIT_INST_ANY"] / wc, 3),
            "lds_bank_conflict_frac": round(v["SQ_LDS_BANK_CONFLICT"] / wc, 4),
            # SQ_INSTS_VALU COUNTS the MFMA instructions themselves (8192^3 GEMM, whose K loop has no other VALU: 1.09,
            # profiles/r04_gemm8192_pmc_insts_valu_includes_mfma.txt): the second figure is the non-matrix VALU per MFMA
            "valu_per_mfma": round(v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], 2) if v["SQ_INSTS_MFMA"] else None,
            "other_valu_per_mfma": round(v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"] - 1.0, 2) if v["SQ_INSTS_MFMA"] else None,
        }
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                         "SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python3 bench.py "
                         "--steps 1 --warmup 0 (eager, one lane)", "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
