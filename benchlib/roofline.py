"""Roofline objects of the bench line (bench.py): every figure is recomputable from the fields it carries.  No oracle import here (bench.py's
cpu_baseline / verify legs are the only users of oracle/)."""
import json
import os

import numpy as np  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {
    # BASELINE.json configs[1] / configs[2]
    "c2": dict(w=1920, h=1080, n=2000, levels=3, name="C2 synthetic 1080p@30fps, 2000 KLT tracks, 3 pyramid levels"),
    "c3": dict(w=3840, h=2160, n=5000, levels=4, name="C3 synthetic 4K@30fps, 5000 KLT tracks, 4 pyramid levels"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
CLOCK_GHZ = 2.4
N_SIMD = 256 * 4


def newest_profile(kind):
    """Path of the newest round's profiles/rNN_<kind> (None when there is none)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{kind}")))
    return files[-1] if files else None


def valu_peak():
    """Issue peak of the HALF-RATE opcode class the LK kernels are made of, in T lane-instructions/s: 1024 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3, the
    architectural rate of a 4-cycle wave64 instruction, confirmed (13.5-15 lanes/clk sustained by single-opcode loops) by the committed micro-benchmark
    (tools/ubench/valu_rate.hip -> the newest profiles/rNN_valu_rate.json).  This is the denominator of `frac_of_class_peak`, NOT of `frac`."""
    path = newest_profile("valu_rate.json")
    lanes, src = 16.0, "assumed 16 lanes/clk/SIMD (profiles/r01_lk_sq_pmc.md); micro-benchmark file missing"
    try:
        j = json.load(open(path))
        lanes = float(j["summary"]["int_valu_lanes_per_clk_per_simd"])
        src = f"{os.path.relpath(path, ROOT)} (tools/ubench/valu_rate.hip)"
    except Exception:
        pass
    return N_SIMD * lanes * CLOCK_GHZ * 1e9 / 1e12, lanes, src


MFMA_F64_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: dense f64 matrix peak (v_mfma_f64_16x16x4_f64: 2048 flop / 64 cycles / SIMD)


def lk_valu_model():
    """Wave instructions the fine-stage kernel issues as a function of its in-kernel counters (template set-ups, Newton iterations), fitted
    against rocprofv3 SQ_INSTS_VALU passes at different iteration counts (tools/pmc_lk_calib.sh -> profiles/rNN_lk_valu_model.json, which also holds
    the check run and the tolerance; tools/collect_profiles.sh re-fits it first, so a profile set and its model belong to the same kernel build).
    The newest round's file is used.  Returns None when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_lk_valu_model.json")))
    if not files:
        return None
    rel = os.path.relpath(files[-1], ROOT)
    try:
        j = json.load(open(files[-1]))
        return dict(per_setup=float(j["wave_instr_per_setup"]), per_iter=float(j["wave_instr_per_newton_iter"]), tolerance=float(j["tolerance"]),
                    kernel=j["kernel"], source=f"{rel} (tools/pmc_lk_calib.sh: SQ_INSTS_VALU fitted over runs with different iteration counts)",
                    coarse=j.get("coarse"))
    except Exception:
        return None


def valu_rates():
    """opcode -> measured lanes / clk / SIMD (best over 1-4 waves per SIMD, 16 independent chains), from the newest profiles/rNN_valu_rate.json"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_valu_rate.json")))
    if not files:
        return {}, None
    best = {}
    for r in json.load(open(files[-1]))["results"]:
        if r["chains"] > 1:
            k = r["inst"]
            best[k] = max(best.get(k, 0.0), r["lanes_per_ns_per_simd"] / CLOCK_GHZ)
    return best, os.path.relpath(files[-1], ROOT)


_RATE_ALIAS = {  # ISA spelling (tools/isa_mix.py) -> name in the micro-benchmark
    "v_dot2c_i32_i16": "v_dot2c_i32_i16 (VOP2)", "v_dot2c_i32_i16_dpp": "v_dot2c_i32_i16 row_shl:1 (DPP)", "v_add_u32_dpp": "v_add_u32 row_shr:1 (DPP)",
    "v_sub_u32_dpp": "v_add_u32 row_shr:1 (DPP)", "v_mov_b32_dpp": "v_mov_b32 row_shr:1 (DPP)", "v_mul_i32_i24_sdwa": "v_mul_i32_i24 (SDWA)",
    "v_subrev_u32": "v_sub_u32", "v_cndmask_b32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)", "v_cmp_lt_i32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)",
    "v_cmp_gt_i32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)", "v_cmp_le_i32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)", "v_cmp_ge_i32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)",
    "v_readlane_b32": "v_readlane_b32 + v_writelane_b32 (pair)", "v_writelane_b32": "v_readlane_b32 + v_writelane_b32 (pair)", "v_pk_add_u16": "v_pk_add_u16",
    "v_pk_mad_u16": "v_pk_mad_u16", "v_pk_sub_i16": "v_pk_sub_i16", "v_fmac_f64": "v_fma_f64", "v_pk_mul_f32": "v_pk_fma_f32", "v_pk_add_f32": "v_pk_fma_f32",
    "v_fmac_f32": "v_fma_f32", "v_mul_lo_u32": "v_mul_lo_u32", "v_min_i32": "v_min_i32", "v_max_i32": "v_max_i32"}


def valu_mix(kernel, setups, iters, per_setup=None, per_iter=None):
    """Instruction mix of one launch of an LK kernel: the static opcode histograms of its set-up and Newton-iteration blocks (profiles/rNN_lk_isa_mix.json,
    tools/isa_mix.py) weighted by the LIVE set-up / iteration counters (x the fitted wave instructions per set-up / iteration when a PMC fit exists,
    else the static block sizes), every opcode priced with its measured issue rate (profiles/rNN_valu_rate.json).  Returns None without the files."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_lk_isa_mix.json")))
    rates, rsrc = valu_rates()
    if not files or not rates:
        return None
    k = json.load(open(files[-1]))["kernels"].get(kernel.replace(" ", ""))
    if not k:
        return None
    hs, hi = k["setup"]["opcodes"], k["iteration"]["opcodes"]
    ns, ni = float(sum(hs.values())), float(sum(hi.values()))
    ws = (per_setup if per_setup else ns / (2 if kernel.startswith(("k_lk_o", "k_lk_q")) else 1)) * setups  # (the coarse kernels inline both directions: two static copies)
    wi = (per_iter if per_iter else ni / (2 if kernel.startswith(("k_lk_o", "k_lk_q")) else 1)) * iters
    tot = ws + wi
    if tot <= 0:
        return None
    frac = {}
    for h, n, w in ((hs, ns, ws), (hi, ni, wi)):
        for op, c in h.items():
            frac[op] = frac.get(op, 0.0) + (c / n) * (w / tot)
    cls = dict(full_rate=0.0, half_rate=0.0, slow=0.0, unmeasured=0.0)
    cyc, per_op = 0.0, []
    for op, f in frac.items():
        r = rates.get(_RATE_ALIAS.get(op, op))
        if r is None:
            cls["unmeasured"] += f
            r_eff = 16.0  # priced like the half-rate class
        else:
            cls["full_rate" if r >= 20.0 else ("half_rate" if r >= 12.0 else "slow")] += f
            # the measurement CLASSIFIES the opcode; the ceiling uses the class's architectural issue rate (a wave64 instruction occupies its SIMD for 2 or
            # 4 cycles = 32 / 16 lanes per clock: the single-opcode loops of the micro-benchmark sustain 23-27 / 13.9-14.5 of it, and a kernel that mixes
            # opcodes and wavefronts can -- and round 4's does -- issue faster than they did), slow opcodes (f64, lane moves) their measured rate
            r_eff = 32.0 if r >= 20.0 else (16.0 if r >= 12.0 else r)
        cyc += f / r_eff
        per_op.append((f / r_eff, op, f, r))
    per_op.sort(reverse=True)
    return dict(full_rate_frac=round(cls["full_rate"], 4), half_rate_frac=round(cls["half_rate"], 4), slow_frac=round(cls["slow"], 4), unmeasured_frac=round(cls["unmeasured"], 4),
                classes="full: measured >= 20 lanes/clk/SIMD (v_add_u32, v_sub_u32, v_and_b32, v_ashrrev_i32, f32 add / mul / fma ...); half: 12-20 (v_dot2*, v_perm, v_mad_i32_i24, "
                        "v_lshl_add, v_pk_*, DPP ...); slow: < 12 (f64, v_cndmask pairs, lane moves); unmeasured opcodes are priced like the half-rate class",
                mix_ceiling_lanes_per_clk_per_simd=round(1.0 / cyc, 2),
                top5_by_issue_cycles=[dict(opcode=op, share_of_instructions=round(f, 4), share_of_issue_cycles=round(c / cyc, 4), lanes_per_clk=(round(r, 1) if r else None))
                                      for c, op, f, r in per_op[:5]],
                setup_share_of_instructions=round(ws / tot, 4),
                source=f"{os.path.relpath(files[-1], ROOT)} (tools/isa_mix.py: static hot-path opcode histograms) x this run's set-up / iteration counters; rates from {rsrc}")


def hbm_traffic_profile(cfg_is_c2):
    """The newest committed PMC collection (profiles/rNN_hbm_traffic.json: FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh) or (None, None)."""
    path = newest_profile("hbm_traffic.json")
    if not cfg_is_c2 or not path:
        return None, None
    try:
        return json.load(open(path)), os.path.relpath(path, ROOT)
    except Exception:
        return None, None


def roofline_of(wl, m, world):
    """roofline object: the dominant kernel (fine-stage LK launch) priced against the bound that really limits it -- VALU instruction issue -- with its
    HBM figure next to it, and one row per other kernel family of the step (time from HIP events inside the library, algorithmic bytes, HBM fraction).
    `frac` is what a reader derives from MI355X_MICROARCH.md alone: issued lane-instructions / launch time / (1024 SIMDs x 32 lanes x 2.4 GHz); the
    class-relative figure (the kernel is made of 4-cycle opcodes: 16 lanes / clk) is `frac_of_class_peak`.  Everything is recomputable from the fields."""
    N, SG, cfg = wl.N, wl.SG, wl.cfg
    prof = m["prof"]
    ms_sum, launches, iters, setups = prof["ms_sum"], prof["launches"], prof["iters"], prof["setups"]
    st_ms, st_n = m["stage_ms"], m["stage_n"]
    wf, wc = 51, 15
    us_fine = 1e3 * ms_sum[2] / max(launches[2], 1)
    # algorithmic gather bytes per launch of session group 0 (SG streams): SURVEY §8d, KLT track solve row: 2 N L [(w+2)^2 + (w+1)^2], L = 1
    bytes_fine = 2 * N * SG * 1 * ((wf + 2) ** 2 + (wf + 1) ** 2)
    achieved = bytes_fine / (us_fine * 1e-6) / 1e9 if us_fine > 0 else 0.0
    it_f = iters[2] / max(launches[2], 1)
    su_f = setups[2] / max(launches[2], 1)
    ops_fine = wf * wf * (47.0 * su_f + 12.0 * it_f)  # SURVEY §8d op model: 47 op/px set-up, 12 op/px per Newton iteration
    # the kernels the library says it launched (vh_profile_lk_routes): nobody mirrors vh_lk_route's thresholds
    names = m["lk_kernels"]
    names_src = "vh_profile_lk_routes (the launcher's own routing decision)"
    fine_kernel = names[2]
    # HBM bytes of that kernel are NOT measured by this run: they come from the PMC passes committed under profiles/ (collected at the stream count
    # stored in the file, scaled linearly); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    traffic, sq_util, tsrc = None, None, None
    tj, tname = hbm_traffic_profile(cfg is CONFIGS["c2"])
    if tj is not None:
        k = tj.get(fine_kernel)
        if k is not None:
            traffic = int((2 * k["fetch_kib"] + k["write_kib"]) * 1024 * SG / tj["streams"])
        sq_util = tj.get("sq_valu_issue_utilisation", {}).get(fine_kernel)
        tsrc = f"{tname} (rocprofv3 --pmc pass of tools/collect_profiles.sh, scaled to {SG} streams; not measured in this run)"
    class_tops, lanes, class_src = valu_peak()
    abs_peak = N_SIMD * 32 * CLOCK_GHZ * 1e9 / 1e12  # MI355X_MICROARCH.md: SIMD-32, a full-rate wave64 instruction issues in 2 cycles = 32 lanes / clk / SIMD
    model_tops = ops_fine / (us_fine * 1e-6) / 1e12 if us_fine > 0 else 0.0
    # VALU instructions issued per launch: LIVE from this run's in-kernel counters through the calibrated per-set-up / per-iteration costs
    issued, isrc, tol = None, None, None
    vm = lk_valu_model()
    if vm is not None and vm["kernel"] == fine_kernel:
        issued = 64.0 * (vm["per_setup"] * su_f + vm["per_iter"] * it_f)
        isrc, tol = f"64 lanes x ({vm['per_setup']:.1f} x set-ups + {vm['per_iter']:.1f} x Newton iterations) per launch, counters of THIS run; costs from {vm['source']}", vm["tolerance"]
    issued_tops = issued / (us_fine * 1e-6) / 1e12 if issued and us_fine > 0 else None
    mix = valu_mix(fine_kernel, su_f, it_f, vm["per_setup"] if vm and vm["kernel"] == fine_kernel else None, vm["per_iter"] if vm and vm["kernel"] == fine_kernel else None)

    # ---- the other kernel families of a step: live HIP-event time + algorithmic bytes (ROI sizes read back from the device after the run) ----
    def us(stage):
        return 1e3 * st_ms[stage] / max(st_n[stage], 1) if st_n[stage] else None

    steps_prof = max(launches[2], 1)
    roi = m["rois"]  # [SG, 4] x0 x1 y0 y1 of the last frame
    rw, rh = (roi[:, 1] - roi[:, 0]).astype(float), (roi[:, 3] - roi[:, 2]).astype(float)
    roi_px = float((rw * rh).sum())
    lc = wl.lk_levels if getattr(wl, "lk_levels", None) is not None else (cfg["levels"] - 1 if wl.params == "baseline" else 4)
    sw, sh = round(cfg["w"] * 0.25), round(cfg["h"] * 0.25)
    rows = []

    def row(kernel, stage, alg_bytes, per_step_launches, note):
        t = us(stage)
        if t is None:
            return
        t_step = t * st_n[stage] / steps_prof  # stage time per step (a stage can launch more than once per step)
        gbs = alg_bytes / (t_step * 1e-6) / 1e9 if t_step > 0 else 0.0
        rows.append(dict(kernel=kernel, us_per_step=round(t_step, 2), launches_per_step=round(st_n[stage] / steps_prof, 2), alg_bytes_per_step=int(alg_bytes),
                         hbm_gbs=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4), bytes=note))

    gather_c = 2 * N * SG * (lc + 1) * ((wc + 2) ** 2 + (wc + 1) ** 2)
    cm = (vm or {}).get("coarse") if vm else None
    for stg, nm in ((0, names[0] + " (stage 1: quarter-scale image)"), (1, names[1] + " (stage 2: ROI)")):
        ck = names[stg]
        t = 1e3 * ms_sum[stg] / max(launches[stg], 1)
        r_ = dict(kernel=nm, us_per_step=round(t, 2), launches_per_step=1.0, alg_bytes_per_step=int(gather_c),
                  hbm_gbs=round(gather_c / (t * 1e-6) / 1e9, 1) if t > 0 else None, hbm_frac=round(gather_c / (t * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if t > 0 else None,
                  bytes="2 N L [(w+2)^2 + (w+1)^2] gather bytes, w = 15, L = pyramid levels; VALU bound like the fine stage")
        su_c, it_c = setups[stg] / max(launches[stg], 1), iters[stg] / max(launches[stg], 1)
        r_["setups_per_launch"], r_["newton_iters_per_launch"] = int(su_c), int(it_c)
        if cm and cm.get("kernel") == ck and t > 0:
            # LIVE: this run's in-kernel counters through the fitted per-set-up / per-iteration wave-instruction costs (per TRACK counters: the idle lanes of
            # a wavefront whose tracks need different iteration counts are inside the fitted per-iteration cost)
            lane_instr = 64.0 * (cm["wave_instr_per_setup"] * su_c + cm["wave_instr_per_newton_iter"] * it_c)
            r_["valu_frac"] = round(lane_instr / (t * 1e-6) / 1e12 / abs_peak, 4)
            r_["valu_frac_of_class_peak"] = round(lane_instr / (t * 1e-6) / 1e12 / class_tops, 4)
            r_["issued_ginstr_per_launch"] = round(lane_instr / 1e9, 3)
            r_["valu_source"] = f"64 lanes x ({cm['wave_instr_per_setup']:.1f} x set-ups + {cm['wave_instr_per_newton_iter']:.1f} x Newton iterations), counters of THIS run; costs fitted by tools/pmc_lk_calib.sh (tolerance {cm.get('tolerance')}); valu_frac against 32 lanes/clk/SIMD, valu_frac_of_class_peak against 16"
            r_["mix"] = valu_mix(ck, su_c, it_c, cm["wave_instr_per_setup"], cm["wave_instr_per_newton_iter"])
        rows.append(r_)
    row("k_roi_warp (stage 3: float32 affine map + 5-bit bilinear remap of the ROI)", 3, 2.0 * roi_px, 1, "ROI read + ROI written (sum over the streams' ROIs of the last frame)")
    pyr_bytes = SG * sw * sh * sum(4.0 ** -l * 1.25 for l in range(lc)) + 2.0 * roi_px * sum(4.0 ** -l * 1.25 for l in range(lc))
    row("k_pyr_down + k_pyr_pad (quarter-scale pyramid of the new frame; ROI pyramids of both frames)", 4, pyr_bytes, 2 * lc,
        "level l reads 4^-l and writes 4^-(l+1) of its image: new quarter-scale frame + the two ROI crops")
    row("k_ransac_fused (2 x estimateAffine2D)", 5, 2 * 2 * 16.0 * N * SG, 2, "pairs read once per call (16 B each): latency / VALU bound, the byte figure is nominal")
    row("k_resize_quarter", 6, SG * (cfg["w"] * cfg["h"] / 16.0) * 2, 1, "1/16 of the pixels read, as many written")
    row("k_sess_frame (bookkeeping + fused LM pose + records)", 7, SG * N * (8 + 24 + 2 + 4) * 1.0, 1, "track state read (p, p3, masks, ids): latency bound (all LM iterations in one workgroup)")
    accounted = us_fine + sum(r["us_per_step"] for r in rows)
    hbm = dict(achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), alg_bytes_per_launch=bytes_fine,
               traffic=traffic, traffic_source=tsrc, note="algorithmic gather bytes 2 N [(51+2)^2 + (51+1)^2] per stream over the launch time: far below the HBM roof, the kernel is not memory bound")
    # HBM traffic of the WHOLE step: the PMC bytes of every kernel of a step (committed collection, scaled to this run's streams) over this run's step time
    step_hbm = None
    if tj is not None and tj.get("step_total_kib") and m.get("step_us"):
        b = float(tj["step_total_kib"]) * 1024.0 * SG * int(getattr(wl, "G", 1)) / tj["streams"]  # (every session group of the step)
        step_hbm = dict(bytes_per_step=int(b), gbs=round(b / (m["step_us"] * 1e-6) / 1e9, 1), frac_of_hbm_peak=round(b / (m["step_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        step_us=round(m["step_us"], 1), source=f"{tname}: sum over every kernel of a step of (2 x FETCH_SIZE + WRITE_SIZE) per launch x launches per step, scaled to {SG * int(getattr(wl, 'G', 1))} streams; step time of THIS run")
    out = dict(bound="valu", kernel=fine_kernel + " (fine stage: 51x51 window, level 0, fwd+bwd)", kernel_source=names_src,
               achieved=round(issued_tops, 3) if issued_tops else None, peak=round(abs_peak, 1), unit="T lane-instr/s",
               frac=round(issued_tops / abs_peak, 4) if issued_tops else None,
               peak_source="MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz (a full-rate wave64 VALU instruction issues in 2 cycles = 32 lanes / clk / SIMD)",
               frac_of_class_peak=round(issued_tops / class_tops, 4) if issued_tops else None, peak_class=round(class_tops, 1),
               peak_class_lanes_per_clk_per_simd=lanes, peak_class_source=class_src,
               frac_of_mix_ceiling=(round(issued_tops / (N_SIMD * mix["mix_ceiling_lanes_per_clk_per_simd"] * CLOCK_GHZ * 1e9 / 1e12), 4) if issued_tops and mix else None),
               mix=mix, us_per_launch=round(us_fine, 2), launches_per_step=int(getattr(wl, "G", 1)), streams_per_launch=int(SG),
               concurrency=(None if getattr(wl, "G", 1) == 1 else f"{wl.G} sessions on {wl.G} HIP streams: the launch durations (HIP events on session 0's stream, and rocprofv3 "
                            "alike) are measured while the other session's kernels share the chip"),
               issued_ginstr_per_launch=round(issued / 1e9, 3) if issued else None, issued_source=isrc, issued_model_tolerance=tol,
               setups_per_launch=int(su_f), newton_iters_per_launch=int(it_f), simds=N_SIMD, clock_ghz=CLOCK_GHZ,
               note="track solve is VALU-issue bound (SURVEY §8d): frac = issued lane-instructions / launch time / (1024 SIMDs x 32 lanes x 2.4 GHz), the guide's SIMD-32 "
                    "issue rate, which only full-rate opcodes approach (measured 23-27 lanes/clk); the kernel is made of 4-cycle opcodes (mix.half_rate_frac: v_dot2*, "
                    "v_perm, v_lshl_add, v_mul_lo ...; 16 lanes/clk architectural, 13.5-15 measured): frac_of_class_peak prices the same lane-instructions against "
                    "1024 x 16 x 2.4 GHz, frac_of_mix_ceiling against the rate a perfect scheduler reaches with THIS opcode mix",
               # the contract's HBM view of the same kernel (secondary: achieved GB/s of its algorithmic bytes, PMC traffic)
               hbm=hbm, traffic=traffic, step_hbm=step_hbm,
               op_model=dict(gops_per_launch=round(ops_fine / 1e9, 4), tops=round(model_tops, 3),
                             note="SURVEY §8d counts 47 op/px per set-up and 12 op/px per Newton iteration for a straightforward kernel; this kernel issues fewer "
                                  "instructions for the same integers (packed int16 dot products) -- frac prices issued instructions"),
               newton_iters_per_track_dir=round(it_f / (2 * N * SG), 2), sq_valu_issue_utilisation=sq_util,
               lk_kernels=list(names),
               lk_us_per_launch=[round(1e3 * ms_sum[k] / max(launches[k], 1), 2) for k in range(3)],
               lk_newton_iters_per_setup=[round(iters[k] / max(setups[k], 1), 2) for k in range(3)],
               lk_setups_per_track=[round(setups[k] / max(launches[k], 1) / (N * SG), 2) for k in range(3)],
               kernels=rows, step_us_accounted=round(accounted, 1), roi_mean_px=[round(float(rw.mean()), 1), round(float(rh.mean()), 1)])
    return out


def headline_hbm(cfg, fps):
    """SURVEY §8d 'Headline KLT number': (B_img + 21 N) bytes per tracked frame x frames/s against the HBM peak."""
    L_ = cfg["levels"]
    b_img = cfg["w"] * cfg["h"] * (1.0 + 2.0 * sum(4.0 ** -l for l in range(1, L_)))
    per_frame = b_img + 21.0 * cfg["n"]
    gbs = per_frame * fps / 1e9
    return dict(bytes_per_frame=int(per_frame), achieved_gbs=round(gbs, 2), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 5),
                note="image-stage algorithmic bytes only: the step is VALU / latency bound, nowhere near HBM bound (as SURVEY §8d predicted)")

