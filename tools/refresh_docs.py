#!/usr/bin/env python3
"""Rewrites the numeric blocks of profiles/README.md, DESIGN.md (section 5's round-5 results table) and README.md from the committed
profile set (profiles/r05_*): run after `tools/pmc_bench.sh r05 variants` + copying gpurun_out/r05_* into profiles/. The blocks sit
between <!-- r05:begin NAME --> / <!-- r05:end NAME --> markers (inserted on first use around the existing text)."""
import csv, json, re, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda f: os.path.join(R, "profiles", f)
J = lambda f: json.load(open(P(f)))
d = J("r05_bench.json"); r = d["roofline"]; k = r["all_conv_kernels"]; pm = J("r05_pmc_summary.json")
p = pm["kernels"]; sha = pm["git_sha"]; b4 = d["value_batch4"]; c = d["c5_stress"]
g = lambda n: J("r05_bench_%s.json" % n)["value"]; m = lambda n: J("r05_bench_%s.json" % n)["ms_per_step"]
t = J("r05_train_bench.json"); u = J("r05_bench_under_rocprof.json")
kk = lambda n: "%.4f (%.3f)" % (k[n]["ms_per_frame"], k[n]["frac_of_its_peak"])
ks = {}
for row in csv.DictReader(open(P("r05_bench_kernel_stats.csv"))):
    n = re.sub(r"\(anonymous namespace\)::", "", row["Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
    ks[n] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
steps_prof = u["steps"] + u["warmup"] + 10          # timed + warm-up + the roofline / hbm passes of that run (per-step counts below are from the bench's own line)
pmv = lambda n: (p[n]["hbm_bytes_per_launch_corrected"] / 1e9, p[n]["mfma_busy_frac_of_simd_cycles"])
cs = r["clock_state"]

rows = []
rows.append("| `r05_bench.json` | `python bench.py` | **%.1f frames/s** (%.2f ms per 48-frame step); `results_digest` %s = round 4's and the first half of this round's (side streams change no detection), %d steps compared with the single-stream pass; roofline (per-launch pass: one stream, side streams off): `%s` **%.1f TF = %.3f**, launch %.1f µs, traffic %.3f GB, `clock_state` %s MHz; `all_conv_kernels` ms/frame (fraction of its ceiling): window `<128>` %s, `<64,256>` %s, `<16,256>` %s, `tile_conv_f16p_kernel<128,128>` %s, row-wave `<32,2>` %s, `<64,2>` %s, `<128,2>` %s, level 1 %.4f + %.4f + %.4f; `hbm_stages` µs/frame: voxelizer %.1f, rulebooks %.1f + %.1f, level ≤ 16 convs %.1f; extras: `value_host_input` %.1f (%.3f), `value_two_stage` **%.1f**, `value_two_stage_anchor` **%.1f** (661.7 before `cpd_nms_batch_first`), `value_batch4` **%.1f** / %.1f two in flight (first half of the round: 740.0 / 995.9), `latency_1frame_ms` **%.2f** (3.04), `module_api` **%.1f** (907.1), `value_fp32_mfma` %.1f, `train_step` **%.2f ms** (child process; 9.88 in-process before), 8 frames %.1f ms = %.1f frames/s; `c5_stress` (16 frames per call; 160k / 1M points): voxelize + index %.1f / %.1f µs/frame, `rulebook_subm` %.2f / %.2f TB/s, SubM 16 / 32 / 64: %.2f / %.2f / %.2f TB/s; `cpu_baseline` %.3f frames/s (%s), %.4f on one thread |" % (
    d["value"], d["ms_per_step"], d["results_digest"]["timed_steps"][0], d["results_digest"]["steps_compared"], r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r["traffic"] / 1e9, cs["sclk_MHz_median"],
    kk("window_conv_f16p_kernel<128>"), kk("window_conv_f16p_kernel<64,256>"), kk("window_conv_f16p_kernel<16,256>"), kk("tile_conv_f16p_kernel<128,128>"), kk("rowwave_conv_f16pe_kernel<32,2>"), kk("rowwave_conv_f16pe_kernel<64,2>"), kk("rowwave_conv_f16pe_kernel<128,2>"),
    k["gather_conv_h16_kernel<2,1>"]["ms_per_frame"], k["gather_conv_h16_kernel<2,2>"]["ms_per_frame"], k["gather_conv_kernel<2,1,false>"]["ms_per_frame"],
    d["hbm_stages"]["voxelize+mean_vfe"]["us_per_frame"], d["hbm_stages"]["rulebook_subm"]["us_per_frame"], d["hbm_stages"]["rulebook_conv"]["us_per_frame"], d["hbm_stages"]["sparse_conv_c<=16"]["us_per_frame"],
    d["value_host_input"]["value"], d["value_host_input"]["ratio_to_value"], d["value_two_stage"]["value"], d["value_two_stage_anchor"]["value"], b4["value"], b4["two_batches_in_flight"]["value"], d["latency_1frame_ms"], d["module_api"]["value"], d["value_fp32_mfma"]["value"],
    d["train_step"]["ms_per_step"], d["train_step_8frames"]["ms_per_step"], d["train_step_8frames"]["frames_per_s"],
    c["160k_points"]["voxelize+mean_vfe+index"]["us_per_frame"], c["1000k_points"]["voxelize+mean_vfe+index"]["us_per_frame"], c["160k_points"]["rulebook_subm"]["TBps"], c["1000k_points"]["rulebook_subm"]["TBps"],
    c["160k_points"]["subm_conv_16"]["TBps"], c["160k_points"]["subm_conv_32"]["TBps"], c["160k_points"]["subm_conv_64"]["TBps"],
    d["cpu_baseline"]["value"], re.search(r"\): (.*)$", d["cpu_baseline"]["sample"]).group(1), d["cpu_baseline"]["one_thread"]["value"]))
A = lambda n: ks[n][0]
tot_ms = sum(a * n for a, n in ks.values()) / 1e3
rows.append("| `r05_bench_kernel_stats.csv`, `r05_bench_under_rocprof.json` | `rocprofv3 --kernel-trace --stats … -- python bench.py --streams 1 --no-cpu-baseline --no-extras` | %.0f frames/s under the tracer; `window_conv_f16p_kernel<128,128>` **%.1f µs** × 11 per step (HIP events in the headline run %.1f), `<64,256>` %.1f × 2, `window_conv_f16p16_kernel<256>` %.1f, `tile_conv_f16p_kernel` %.1f × 3, row-wave `<128,2>` %.1f × 6, `<64,2>` %.1f × 5, `<32,2>` %.1f × 4. With the index chain on its side stream the per-kernel times no longer add up to the step: `rulebook_chunk_kernel` %.1f µs × 6 and `order_rows_kernel` %.1f × 3 are stretched by the convolutions they run beside (200.5 / 54.6 µs alone, first half of the round) — the time they take is hidden, not spent |" % (
    u["value"], A("window_conv_f16p_kernel<128, 128>"), r["avg_launch_us"], A("window_conv_f16p_kernel<64, 256>"), A("window_conv_f16p16_kernel<256>"), A("tile_conv_f16p_kernel"),
    A("rowwave_conv_f16pe_kernel<128, 2>"), A("rowwave_conv_f16pe_kernel<64, 2>"), A("rowwave_conv_f16pe_kernel<32, 2>"), A("rulebook_chunk_kernel"), A("order_rows_kernel<1024, 4>")))
rows.append("| `r05_pmc_summary.json` | three `--pmc` passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) of `python bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --no-extras` | window `<128>`: %.3f GB per launch, MFMA busy **%.3f**; `<64,256>` %.2f GB, %.2f; head tile %.2f GB, %.2f; row-wave `<32,2>` / `<64,2>` / `<128,2>` %.2f / %.2f / %.2f GB, %.2f / %.2f / %.2f (unchanged: no conv kernel changed in the second half of the round; the source hash moved with `center_targets.hip` and `cpd_nms_batch_first`) |" % (
    pmv("window_conv_f16p_kernel<128>") + pmv("window_conv_f16p_kernel<64,256>") + pmv("window_conv_f16p16_kernel<256>") +
    (pmv("rowwave_conv_f16pe_kernel<32,2>")[0], pmv("rowwave_conv_f16pe_kernel<64,2>")[0], pmv("rowwave_conv_f16pe_kernel<128,2>")[0], pmv("rowwave_conv_f16pe_kernel<32,2>")[1], pmv("rowwave_conv_f16pe_kernel<64,2>")[1], pmv("rowwave_conv_f16pe_kernel<128,2>")[1])))
rows.append("| `r05_bench_fp32_dense_maps.json` | `--dense-pairs 0` (same box, same run) | %.1f: the round-4 dense half under this round's harness — pairs +%.1f %% |" % (g("fp32_dense_maps"), 100 * (d["value"] / g("fp32_dense_maps") - 1)))
rows.append("| `r05_bench_streams1.json`, `…streams3.json` | `--streams 1 | 3` | %.1f / %.1f (two: %.1f) |" % (g("streams1"), g("streams3"), d["value"]))
rows.append("| `r05_bench_bf16x3.json`, `…f32.json`, `…canonical_rows.json`, `…device_results.json`, `…modules.json` | one flag each | %.1f / %.1f / %.1f / %.1f / %.1f |" % tuple(g(n) for n in ("bf16x3", "f32", "canonical_rows", "device_results", "modules")))
rows.append("| `r05_bench_16frames.json`, `…4frames.json`, `…4frames_2streams.json`, `…1frame.json` | `--frames 16`; `--frames 4 --streams 1`; `--frames 4`; `--frames 1 --streams 1` | %.1f; **%.1f** (%.2f ms; 737.5 before the side streams); **%.1f** (1009.0); **%.1f (%.2f ms; 3.02)** — DESIGN §5j |" % (g("16frames"), g("4frames"), m("4frames"), g("4frames_2streams"), g("1frame"), m("1frame")))
rows.append("| `r05_train_bench.json`, `r05_train_kernel_stats.csv`, `r05_train_bench_under_rocprof.json` | `python bench.py --mode train --steps 40 --warmup 10` (sets `GPU_MAX_HW_QUEUES=8`) | **%.2f ms/step = %.1f train frames/s** (9.62 / 104.0 in the first half of the round: index chain on its own stream, `cpd_center_targets`, eight hardware queues — DESIGN §5a); 300-step same-box pairs: 9.74 / 9.90 → 8.78–9.33 |" % (t["ms_per_step"], t["value"]))
rows.append('| `r05_rowwave_pmc.json` | `tools/pmc_rowwave.sh r05` (13 `--pmc` passes of the default bench, one counter block each) | the row-wave kernels at the end of the round (unchanged since round 4: the reference set for the §8.1 discussion) — `<32,2>` / `<64,2>` / `<128,2>`: 917 / 1005 / 1185 µs, MFMA busy 0.23 / 0.37 / 0.46, waves per CU 25.7 / 17.7 / 10.3, TD busy **0.96 / 0.89 / 0.72**, TA busy 0.80 / 0.64 / 0.44, L1 hit 0.54 / 0.53 / 0.53 (L1 latency 62 / 46 / 36 cycles per access), L2 hit 0.79 / 0.80 / 0.78 (L2 read latency 305 / 259 / 233 cycles), wave cycles waiting 0.47 / 0.44 / 0.30, issue-stalled 0.34 / 0.33 / 0.44, 2.91 / 2.77 / 2.85 GB per launch |')
rows.append("| `r05_band_order_probe.txt` | `FRAMES=48 ORDER_SET=band python tools/order_probe.py f16x2` | the row-wave kernels on canonical, band-major ((b, y-band, z, y, x), bands of 8 … 64 lines) and pattern-sorted orders of the three levels — and on a SYNTHETIC rulebook of perfect locality (`LOCAL`): 932.7 → 910.7, 1140.5 → 1110.2, 1534.6 → 1494.7 µs. L2 misses are worth 2.5 % of these kernels: DESIGN §8.1 |")
rows.append("| `r05_local_pmc.txt` | `tools/local_pmc.sh` (six `--pmc` passes of `tools/local_probe_one.py` per level and variant) | the counters behind the row above: real vs perfect-locality rulebook, 32 ch: L2 hit 0.826 → 0.867, fabric fetch 1793 → 1233 MB, TD busy 0.94 / 0.93, L1 accesses and L1 → L2 reads identical, **1015.5 vs 1026.8 µs**; 128 ch: 0.828 → 0.880, 2454 → 1573 MB, 1791.3 vs 1783.1 µs |")
rows.append("| `r05_1frame_timeline.txt` | `rocprofv3 --kernel-trace … bench.py --frames 1 --streams 1`, then `tools/trace_gaps.py … --between select_boxes_kernel 30 60` + one step kernel by kernel (queue, start µs, duration µs) | the one-frame step under the tracer (3.29 ms; 2.7–2.8 untraced): main queue 2.42 ms of kernels per step, the index queue 0.57 ms running beside the convolutions of the stage before; what is left on the main queue: 17 × `split_finish`, the decode / NMS tail, the shared conv on the table path |")
table = "| file | command | what to read |\n|---|---|---|\n" + "\n".join(rows) + "\n"


def put(path, name, text, find_a=None, find_b=None):
    s = open(path).read()
    a, b = "<!-- r05:begin %s -->\n" % name, "<!-- r05:end %s -->\n" % name
    if a in s:
        i, j = s.index(a) + len(a), s.index(b)
        s = s[:i] + text + s[j:]
    else:
        i, j = s.index(find_a), s.index(find_b)
        s = s[:i] + a + text + b + s[j:]
    open(path, "w").write(s)


put(os.path.join(R, "profiles", "README.md"), "table", table + "\n", "| file | command | what to read |", "Same-box pairs of the second half of the round")
s = open(os.path.join(R, "profiles", "README.md")).read()
s = re.sub(r"`CPD_GIT_SHA=[0-9a-f]+ tools/pmc_bench.sh r05 variants`", "`CPD_GIT_SHA=%s tools/pmc_bench.sh r05 variants`" % sha, s, count=1)
open(os.path.join(R, "profiles", "README.md"), "w").write(s)

design = '''Results on MI355X, END of round 5 (`profiles/r05_*`: one `CPD_GIT_SHA=%s tools/pmc_bench.sh r05 variants` run on one box — this
round's boxes printed 1117–1254; the driver's round-4 box was 4 %% below that round's profile box — quote the range; in brackets: the
first half of the round, before the side streams of §5j, on ITS profile box):

| config | frames/s | ms |
|---|---|---|
| **48 frames/step, 2 batches in flight (bench.py default)** | **%.1f** (window kernel %.1f TF = %.3f; [1200.9]; round 4 on its profile box: 1183.8) | %.2f per step = %.3f per frame |
| same run, `--dense-pairs 0` (fp32 dense maps: the round-4 dense half) | %.1f (pairs: +%.1f %%; other boxes +2.0 %%, +2.2 %%, +2.6 %%, +3.1 %%) | |
| `--streams 1` / `--streams 3` | %.1f (%.1f under rocprofv3) / %.1f | |
| `value_host_input` | %.1f (%.3f of `value`) | |
| `--api modules` / `module_api` | **%.1f / %.1f** [901.5 / 907.1: one result copy per key, levels in tap-pattern order (§5e)] | |
| `--conv-math bf16x3` / `f32` | %.1f / %.1f | |
| `--row-order canonical` | %.1f | |
| `--frames 16` | %.1f | |
| 4 frames/step: one stream / two batches in flight | **%.1f / %.1f** [737.5 / 1009.0] (fp32 dense maps at this batch size) | %.2f per step |
| 1 frame/step | **%.1f** [330.9] | **%.2f** [3.02]; `latency_1frame_ms` in the line %.2f |
| two-stage `VoxelRCNN` (CenterPoint first stage), 16 frames, 497 RoIs per frame | **%.1f** in the line [811.4] (round 4: 757.0) | %.1f per step |
| two-stage `VoxelRCNN` of the dbscan / oyster configs (anchor first stage), 16 frames, 200 RoIs | **%.1f** [661.7: `cpd_nms_batch_first`, §5h] | %.1f per step |
| train step (config 3, 1 frame / 8 frames) | **%.1f** (`--mode train`; %.1f in the line, measured in a child process) / %.1f [104.0 / 178.5] | **%.2f** / %.1f [9.62 / 44.8] |

Per frame at one stream, side streams off (HIP events, `roofline.all_conv_kernels`; ms and fraction of the kernel's own ceiling):
`window_conv_f16p_kernel<128>` %s, `<64,256>` %s, `<16,256>` %s (round 4: 0.0205 / 0.127), `tile_conv_f16p_kernel<128,128>` %s (0.0492 /
0.29); row-wave `<32,2>` %s, `<64,2>` %s, `<128,2>` %s (unchanged kernels; box-dependent ±3 %%); level 1 %.4f + %.4f + %.4f. PMC
(`r05_pmc_summary.json`): window `<128>` MFMA busy %.3f, %.3f GB per launch; `<64,256>` %.3f, %.2f GB; head tile %.2f, %.2f GB
(algorithmic 2.17 GB); row-wave %.2f / %.2f / %.2f, %.2f / %.2f / %.2f GB. rocprofv3 (`r05_bench_kernel_stats.csv`): window `<128,128>`
%.1f µs over 11 launches per step (HIP events in the headline run: %.1f). `hbm_stages` (side streams off): voxelizer %.1f µs/frame,
rulebooks %.1f + %.1f, level ≤ 16 convs %.1f; densify is no longer a stage of its own (persistent map: scatter 3.5 + re-zero 1.3
µs/frame). In the timed region the index stages run beside the convolutions (§5j): under the tracer their kernels stretch
(`rulebook_chunk_kernel` 200 → %.0f µs) and the per-kernel times no longer sum to the step.

''' % (sha, d["value"], r["achieved"], r["frac"], d["ms_per_step"], d["ms_per_step"] / 48,
       g("fp32_dense_maps"), 100 * (d["value"] / g("fp32_dense_maps") - 1), g("streams1"), u["value"], g("streams3"),
       d["value_host_input"]["value"], d["value_host_input"]["ratio_to_value"], g("modules"), d["module_api"]["value"], g("bf16x3"), g("f32"), g("canonical_rows"), g("16frames"),
       g("4frames"), g("4frames_2streams"), m("4frames"), g("1frame"), m("1frame"), d["latency_1frame_ms"],
       d["value_two_stage"]["value"], d["value_two_stage"]["ms_per_step"], d["value_two_stage_anchor"]["value"], d["value_two_stage_anchor"]["ms_per_step"],
       t["value"], d["train_step"]["frames_per_s"], d["train_step_8frames"]["frames_per_s"], t["ms_per_step"], d["train_step_8frames"]["ms_per_step"],
       kk("window_conv_f16p_kernel<128>"), kk("window_conv_f16p_kernel<64,256>"), kk("window_conv_f16p_kernel<16,256>"), kk("tile_conv_f16p_kernel<128,128>"),
       kk("rowwave_conv_f16pe_kernel<32,2>"), kk("rowwave_conv_f16pe_kernel<64,2>"), kk("rowwave_conv_f16pe_kernel<128,2>"),
       k["gather_conv_h16_kernel<2,1>"]["ms_per_frame"], k["gather_conv_h16_kernel<2,2>"]["ms_per_frame"], k["gather_conv_kernel<2,1,false>"]["ms_per_frame"],
       pmv("window_conv_f16p_kernel<128>")[1], pmv("window_conv_f16p_kernel<128>")[0], pmv("window_conv_f16p_kernel<64,256>")[1], pmv("window_conv_f16p_kernel<64,256>")[0],
       pmv("window_conv_f16p16_kernel<256>")[1], pmv("window_conv_f16p16_kernel<256>")[0],
       pmv("rowwave_conv_f16pe_kernel<32,2>")[1], pmv("rowwave_conv_f16pe_kernel<64,2>")[1], pmv("rowwave_conv_f16pe_kernel<128,2>")[1],
       pmv("rowwave_conv_f16pe_kernel<32,2>")[0], pmv("rowwave_conv_f16pe_kernel<64,2>")[0], pmv("rowwave_conv_f16pe_kernel<128,2>")[0],
       A("window_conv_f16p_kernel<128, 128>"), r["avg_launch_us"], d["hbm_stages"]["voxelize+mean_vfe"]["us_per_frame"], d["hbm_stages"]["rulebook_subm"]["us_per_frame"],
       d["hbm_stages"]["rulebook_conv"]["us_per_frame"], d["hbm_stages"]["sparse_conv_c<=16"]["us_per_frame"], A("rulebook_chunk_kernel"))
put(os.path.join(R, "DESIGN.md"), "results", design, "Results on MI355X, END of round 5 (`profiles/r05_*`", "Results on MI355X, END of round 4 (`profiles/r04_*`")

readme = '''Measured on one MI355X (end of round 5; details in DESIGN.md §5, evidence in `profiles/r05_*`): whole path at 48 DISTINCT frames per step
with two batches in flight **1117–1254 frames/s across this round's boxes; %.1f on the profile box** (`profiles/r05_bench.json`:
%.3f ms/frame, digest-checked against a single-stream run; the driver's own round-4 line was 1135.9 where that round's profile box
printed 1183.8 — quote the range, not the best box). On the profile box: %.0f single-stream, %.0f with the clouds starting in pinned host
memory, **%.0f through the drop-in module path**, %.0f–%.0f with fp32-MFMA arithmetic everywhere; **%.0f frames/s** through the two-stage
`VoxelRCNN` engine (497 RoIs per frame; 757 in round 4), **%.0f** through the two-stage engine of the dbscan / oyster configs (anchor-head
first stage; 662 before its proposal NMS stopped at the survivors it keeps), **%.0f frames/s at the reference's eval batch of 4** (%.0f
with two batches in flight; 738 / 1009 before this round's side streams), **%.2f ms for a single frame** (3.02), **train step %.1f ms =
%.0f frames/s per GPU** (9.6 / 104; %.0f at 8 frames per GPU), CPU oracle 0.11–0.16 frames/s on 256 host threads (0.023–0.027 on one). ''' % (
    d["value"], d["ms_per_step"] / 48, g("streams1"), d["value_host_input"]["value"], d["module_api"]["value"], d["value_fp32_mfma"]["value"], g("f32"),
    d["value_two_stage"]["value"], d["value_two_stage_anchor"]["value"], b4["value"], b4["two_batches_in_flight"]["value"], m("1frame"), t["ms_per_step"], t["value"],
    d["train_step_8frames"]["frames_per_s"])
put(os.path.join(R, "README.md"), "measured", readme, "Measured on one MI355X (end of round 5;", "Convolutions with ≥ 32 input channels run on the fp16 matrix pipe")
s = open(os.path.join(R, "README.md")).read()
s = re.sub(r"dominant kernel \(\d+ TFLOP/s fp32-equivalent = 0\.\d+ of the split-fp16 ceiling", "dominant kernel (%.0f TFLOP/s fp32-equivalent = %.2f of the split-fp16 ceiling" % (r["achieved"], r["frac"]), s)
open(os.path.join(R, "README.md"), "w").write(s)
print("refreshed from", sha, "value", round(d["value"], 1))
