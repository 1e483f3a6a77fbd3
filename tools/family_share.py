"""rocprofv3 --kernel-trace --stats of `bench.py --step-only` -> profiles/<name>.json: share of the step's summed kernel time
per kernel family (what bench.py reports next to the roofline entry).  usage: family_share.py <stats dir> <steps> <out.json>"""
import csv, glob, json, re, sys

d, steps, dst = sys.argv[1], float(sys.argv[2]), sys.argv[3]
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
fam = [("split-bf16 GEMM (k_gemm_bf3*, forward / dgrad / wgrad)", r"k_gemm_bf3"), ("fp32 MFMA GEMM (k_gemm<...>: gate, head and cross-network products)", r"k_gemm<"),
       ("fused tower pyramid (k_tower_fwd, k_tower_bwd)", r"k_tower_"), ("BatchNorm / activation (k_bn_act, k_act_bn_bwd, k_act_bwd, k_bn_bwd_apply, k_bn_running)", r"k_bn_|k_act_bwd|k_act_bn_bwd"),
       ("embedding (gather, radix sort, segmented reduce)", r"k_embed|k_rs_|k_segreduce"), ("L2 regularisation (k_l2_*)", r"k_l2_"),
       ("row plan (k_plan_*)", r"k_plan_"), ("row-wise trunk (cross network, linear, group embedding)", r"k_rowwise|k_grp_|k_seg_reduce"),
       ("split-K / bias reductions", r"k_splitk|k_bias_reduce|k_reduce_tiles|k_reduce_tail|k_colsum"),
       ("per-step preparation (mask tables, weight images, transposes, memsets, copies)", r"k_mask_prep|k_prep_wimg|k_transpose|rocclr|k_loss_finish|k_step_total")]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = {"source": f"rocprofv3 --kernel-trace --stats -- python bench.py --step-only --no-graph ({int(steps)} steps)", "sum_of_kernel_us_per_step": round(tot / 1e3 / steps, 1),
       "launches_per_step": round(sum(int(r["Calls"]) for r in rows) / steps, 1), "families": {}}
rest = tot
for name, pat in fam:
    t = sum(float(r["TotalDurationNs"]) for r in rows if re.search(pat, r["Name"]))
    n = sum(int(r["Calls"]) for r in rows if re.search(pat, r["Name"]))
    rest -= t
    out["families"][name] = {"us_per_step": round(t / 1e3 / steps, 1), "share": round(t / tot, 4), "launches_per_step": round(n / steps, 1)}
out["families"]["other"] = {"us_per_step": round(rest / 1e3 / steps, 1), "share": round(rest / tot, 4)}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
