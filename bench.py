#!/usr/bin/env python3
"""bench.py -- images/sec of the FLUX-VAE encode+tag hot path on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch per GPU: fp32 NCHW images resident in
HBM -> vt_encode_tag (HIP encoder + decoder) -> [B,N] logits -> (N>1) one RCCL all-gather of logits.
Workload at N=1: BASELINE.json configs[2] -- batch 16, 1024x1024, 8-head attention decoder, 10k tags,
bf16 MFMA operands / fp32 accumulate.  Weak scaling: the per-GPU batch is fixed as N grows.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the implicit-GEMM MFMA conv):
achieved = its algorithmic FLOPs / its summed launch durations, from HIP events recorded on the launch
stream over the timed region (vt_profile_begin/end in the C ABI; ~140 event records per step -- the same K steps are run
once more without them and reported as `ms_per_step_without_events`).  `cpu_baseline` times the CPU oracle (oracle/, kind "port") on this box's host cores,
rank 0, N=1 only, on a bounded sample; `max_abs_dlogit` / `max_abs_dlatent` compare image 0 of the measured batch with it.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (not the 2:1-sparse headline)
MFMA_FP8_DENSE_PEAK_TFLOPS = 5000.0      # MI355X_MICROARCH.md: ~5 PF dense fp8 (block-scaled f8f6f4 MFMA)
# what the matrix pipe sustains on THIS chip with operands in registers and random data (tools/mfma_peak.hip,
# profiles/r02/mfma_peak_bare_loops.log): v_mfma_f32_16x16x32_bf16 2.04-2.09 PF (2.32 on zeros), v_mfma_scale_f32_32x32x64_f8f6f4 4.1 PF (4.96)
MFMA_BF16_MEASURED_BARE_TFLOPS = 2040.0
MFMA_FP8_MEASURED_BARE_TFLOPS = 4100.0
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5, help="untimed steps first (clock ramp, code objects, allocator)")
    p.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    p.add_argument("--height", type=int, default=1024)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--tags", type=int, default=10000)
    p.add_argument("--encode-only", action="store_true", help="BASELINE configs[1]: encoder only")
    p.add_argument("--bucketed", action="store_true",
                   help="BASELINE configs[3]: same-shape batches drawn from the reference's 512..1024 step-64 aspect buckets")
    p.add_argument("--bucket-batch", type=int, default=8, help="images per same-shape batch in --bucketed mode")
    p.add_argument("--fp8", action="store_true",
                   help="BASELINE configs[4]: the 3x3 convs of the resnet / downsample stack and the attention's Q.K^T / P.V on fp8 (e4m3) "
                        "operands / fp8 MFMA (vt_set_flag 11); opt-in mode, logits within 1e-2 of the CPU reference, latents ~1e-1")
    p.add_argument("--f16", action="store_true",
                   help="fp16 instead of bf16 MFMA operands for the convolutions (vt_set_flag 18): the precision mode -- latents ~6x closer to the "
                        "fp32 reference for a few per cent of the images/s; not BASELINE.json's dtype, never the default line")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-also", action="store_true", help="skip the configs[4] / configs[3] legs attached to the default run's line")
    p.add_argument("--generic-conv", action="store_true", help="A/B: disable the halo-tile 3x3 kernel")
    p.add_argument("--no-occ2", action="store_true", help="A/B: 128-cout convs on the one-workgroup-per-CU tile")
    p.add_argument("--lib", default=None, help="A/B: path of an alternative build of libvae_tagger_hip.so")
    p.add_argument("--flag", action="append", default=[], metavar="N=V", help="A/B: vt_set_flag(N, V) before the run (repeatable)")
    return p.parse_args()


def physical_cores():
    """Physical cores of this host (unique (package, core) pairs of /proc/cpuinfo), capped by the cpuset this process may use."""
    try:
        pairs, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
        n = len(pairs)
    except OSError:
        n = 0
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, avail)) if n else max(1, avail)


def cpu_baseline(x0, tags, flops_target, budget_s=85.0):
    """CPU oracle (torch fp32 restatement of the reference path, oracle/) on this box's host cores, SURVEY.md section 8(d):
    1 warm-up + 3 timed runs per row -- all physical cores at the benchmark shape and at 512^2, one core on a 256^2
    sample (a single core needs minutes for a 1024^2 image) -- bounded by `budget_s`.  x0: the FIRST image of the measured
    batch; the warm-up run's outputs are returned as the parity reference for that image.  Baseline only, never the target."""
    import torch
    from oracle import decoder_ref, encoder_ref
    from vae_tagger_amd import synth
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(tags), seed=1)
    cores = physical_cores()
    t_start = time.perf_counter()

    def run(x):
        with torch.no_grad():
            lat = encoder_ref.vae_wrapper_encode(sd_e, x)
            lg = decoder_ref.attention_decoder_forward(sd_d, lat)
            decoder_ref.get_confidence(lg)
        return lat, lg

    def row(x, threads, label):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        ref = run(x)                                           # warm-up (allocator, oneDNN primitive cache)
        warm = time.perf_counter() - t0
        times = []
        for _ in range(3):
            if time.perf_counter() - t_start + warm > budget_s and times:
                break
            t0 = time.perf_counter()
            run(x)
            times.append(time.perf_counter() - t0)
        h, w = x.shape[-2:]
        scale = encoder_ref.encoder_flops(h, w) / flops_target
        best = min(times)
        return ref, {"threads": threads, "shape": f"{w}x{h}", "timed_runs": len(times), "seconds_best": round(best, 3),
                     "seconds_mean": round(sum(times) / len(times), 3),
                     "images_per_sec_at_bench_shape": round(scale / best, 5), "sample": label}

    H, W = x0.shape[-2:]
    ref, main_row = row(x0, cores, f"image 0 of the measured batch, {W}x{H}, encode+tag")
    rows = [main_row]
    if time.perf_counter() - t_start < budget_s * 0.7:
        rows.append(row(synth.synth_images(1, 512, 512, seed=0), cores, "1 image 512x512 encode+tag, scaled by FLOP ratio")[1])
    if time.perf_counter() - t_start < budget_s * 0.8:
        rows.append(row(synth.synth_images(1, 256, 256, seed=0), 1, "1 image 256x256 encode+tag on ONE core, scaled by FLOP ratio")[1])
    torch.set_num_threads(cores)
    return ref, {"value": main_row["images_per_sec_at_bench_shape"], "unit": "images/sec", "cores": cores, "kind": "port",
                 "sample": f"1 image {W}x{H} encode+tag, fp32 torch CPU oracle, {cores} threads = physical cores, 1 warm-up + "
                           f"{main_row['timed_runs']} timed runs (best {main_row['seconds_best']} s)",
                 "rows": rows, "total_seconds": round(time.perf_counter() - t_start, 1)}


def pmc_traffic(kernel, batch, height, width, fp8=False):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (tools/profile_round.sh: separate
    FETCH_SIZE and WRITE_SIZE passes; both in KB; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md -- round 4's
    tools/fetch_probe.hip: the counter tallies 64 B per request, whole 128-B lines cross the fabric).
    Only valid for the workload the counters were collected on (batch 16 x 1024^2); None otherwise."""
    root = os.path.dirname(os.path.abspath(__file__))
    if (batch, height, width) != (16, 1024, 1024):
        return None, None
    want = kernel.replace(" ", "").rstrip(">")              # (template arguments added since -- e.g. ", false>" -- must not break the match)
    for rnd in ("r04", "r03"):
        path = os.path.join(root, "profiles", rnd, f"pmc_traffic_b16_1024_{'fp8' if fp8 else 'bf16'}.json")
        if not os.path.exists(path):
            continue
        for name, c in json.load(open(path)).items():
            if want in name.replace(" ", "") and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, root)
    return None, None


class SocketPower:
    """hwmon power1_input / freq1_input of the amdgpu cards, polled from a thread while a region runs (the socket sits at its power
    cap under this workload, and that -- not issue slots -- sets the MFMA kernels' rate).  The hwmon tree shows every card of the host; the one
    reported is the card whose power rose most against the idle sample taken before this process touched the GPU.  Absent files: None."""
    def __init__(self):
        import glob
        self.dirs = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.idle = [self._read(d, "power1_input") for d in self.dirs]
        self.rows = []

    @staticmethod
    def _read(d, f):
        try:
            with open(os.path.join(d, f)) as fh:
                return int(fh.read().strip())
        except Exception:
            return None

    def __enter__(self):
        import threading
        self.rows, self._stop = [], False
        def run():
            while not self._stop:
                self.rows.append([(self._read(d, "power1_input"), self._read(d, "freq1_input")) for d in self.dirs])
                time.sleep(0.05)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        self._th.join()

    def summary(self):
        rows = self.rows[len(self.rows) // 3:]                 # the reading is a moving average: drop the part that still holds the previous phase
        if not self.dirs or len(rows) < 3:
            return None
        med = lambda v: (sorted(v)[len(v) // 2] if v else None)
        pw = [med([r[i][0] for r in rows if r[i][0] is not None]) for i in range(len(self.dirs))]
        cand = [i for i in range(len(self.dirs)) if pw[i] is not None]
        if not cand:
            return None
        i = max(cand, key=lambda i: pw[i] - (self.idle[i] or 0))
        cap = self._read(self.dirs[i], "power1_cap")
        fq = med([r[i][1] for r in rows if r[i][1] is not None])
        return {"socket_w_median": round(pw[i] / 1e6, 1), "cap_w": None if cap is None else round(cap / 1e6, 1),
                "smu_sclk_mhz_median": None if fq is None else round(fq / 1e6), "samples": len(rows),
                "source": "hwmon power1_input / freq1_input, 50-ms polling during the K untraced steps (rank 0's view; the card whose power rose most)"}


def main():
    a = parse()
    power = SocketPower()
    if a.lib:
        from vae_tagger_amd import _lib
        _lib.LIB_PATH = os.path.abspath(a.lib)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # rehearsal on a one-GPU box (never used by the driver): VT_BENCH_REHEARSAL=1 puts every rank on GPU 0 and uses gloo for
    # the collectives, so the N > 1 control flow (rank-distinct inputs, all-gather, max-over-ranks timing) runs without RCCL
    rehearsal = os.environ.get("VT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    from vae_tagger_amd import sharding, synth
    from vae_tagger_amd.diffusers_vae_loader import (DiffusersVAEWrapper, get_diffusers_vae_config,
                                                      load_diffusers_vae_from_config)
    from vae_tagger_amd.modules import create_attention_decoder
    from vae_tagger_amd.pipeline import EncodeTagPipeline

    # the mirrors print the reference's construction messages; stdout carries the ONE JSON line only
    import contextlib
    _quiet = contextlib.redirect_stdout(sys.stderr)
    _quiet.__enter__()
    vae = load_diffusers_vae_from_config(get_diffusers_vae_config())
    vae.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    vae_model = DiffusersVAEWrapper(vae).to(dev).eval()
    vae_model.check_finite = False          # no host synchronisation inside the timed steps: the status word is read once after them
    dec = create_attention_decoder(16, a.height // 8, a.width // 8, a.tags,
                                   {"use_spatial_attention": True, "use_self_attention": True,
                                    "use_cross_attention": False, "attention_heads": 8})
    dec.load_state_dict(synth.synth_state_dict(synth.attention_decoder_manifest(a.tags), seed=1), strict=False)
    dec = dec.to(dev).eval()
    pipe = EncodeTagPipeline(vae_model, dec)
    pipe.check_finite = False                 # no host synchronise inside the timed steps: the word is read once after them
    _quiet.__exit__(None, None, None)
    if a.generic_conv:
        pipe.ctx.call("vt_set_flag", 0, 0)
    if a.no_occ2:
        pipe.ctx.call("vt_set_flag", 3, 0)      # one workgroup per CU for every halo conv
    if a.fp8:
        pipe.ctx.call("vt_set_flag", 11, 1)
        if a.encode_only:
            vae._context().call("vt_set_flag", 11, 1)       # (the encode-only leg runs on the VAE mirror's own context)
    if a.f16:
        pipe.set_fp16_operands(True)
        vae._context().call("vt_set_flag", 18, 1)
    for fv in a.flag:
        f, v = fv.split("=")
        pipe.ctx.call("vt_set_flag", int(f), int(v))

    B = a.batch
    counts = [B] * world

    def make_bucket_plan(n_steps, seed=0):
        """reference AspectRatioBucketing(512, 1024, 64) (modules.py:180-222, train_full.sh:13-16): the buckets an image can
        actually be assigned to, a seeded draw of 2*world same-shape batches per step, whole batches placed on ranks by the
        FLOP cost model (conv ~ pixels, attention ~ pixels^2).  Inputs are made resident in HBM here, before any timing."""
        import random
        from vae_tagger_amd.modules import AspectRatioBucketing
        bk = AspectRatioBucketing(512, 1024, 64)
        reach = sorted({bk.bucket_for_ratio(w / h) for w in range(256, 2049, 8) for h in range(256, 2049, 8)})
        rng = random.Random(seed)
        plan = []                                       # per step: [(w, h, n)] and the assignment
        for _ in range(n_steps):
            batches = [(*rng.choice(reach), a.bucket_batch) for _ in range(2 * world)]
            assign, _ = sharding.assign_batches(batches, world)
            plan.append((batches, assign))
        for batches, assign in plan:
            for i in assign[rank]:
                bucket_input(batches[i][0], batches[i][1])
        return plan

    bucket_cache = {}

    def bucket_input(w, h):
        if (w, h) not in bucket_cache:
            bucket_cache[(w, h)] = synth.synth_images(a.bucket_batch, h, w, seed=w * 4096 + h + rank).to(dev)
        return bucket_cache[(w, h)]

    def bucket_step(batches, assign):
        outs = [pipe.logits(bucket_input(batches[i][0], batches[i][1])) for i in assign[rank]]
        local = torch.cat(outs, dim=0) if outs else torch.empty(0, a.tags, device=dev)
        if world == 1:
            return local
        return sharding.all_gather_logits(local, [sum(batches[i][2] for i in assign[r]) for r in range(world)])

    x = None
    if not a.bucketed:
        # synthetic inputs, resident in HBM before the timed region; distinct per rank (global batch = world*B)
        x = synth.synth_images(B, a.height, a.width, seed=1000 + rank).to(dev)

    def plain_step():
        if a.encode_only:
            return vae_model.encode(x)
        logits = pipe.logits(x)
        return sharding.all_gather_logits(logits, counts) if world > 1 else logits

    prof_ctx = pipe.ctx if not a.encode_only else vae._context()

    def timed(step_fns, profile=True):
        """EXACTLY len(step_fns) steps bracketed by barrier + synchronize on both sides; MAX over ranks.  With `profile` the
        library records one hipEvent pair per MFMA / GroupNorm launch on the launch stream (vt_profile_begin/end)."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if profile:
            prof_ctx.call("vt_profile_begin")
        t0 = time.perf_counter()
        out = None
        for f in step_fns:
            out = f()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        prof = None
        if profile:
            n = prof_ctx.lib.vt_profile_num_configs()
            launches = (ctypes.c_longlong * n)()
            tot_ms = (ctypes.c_double * n)()
            tot_fl = (ctypes.c_double * n)()
            names = (ctypes.c_char_p * n)()
            prof_ctx.call("vt_profile_end", n, launches, tot_ms, tot_fl, names)
            prof = (n, list(launches), list(tot_ms), list(tot_fl), [nm.decode() for nm in names])
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out, prof

    def roofline_of(prof, elapsed, traffic_key=None):
        """`roofline` (dominant MFMA kernel) and `hbm_pass` (GroupNorm apply) objects from one profiled region."""
        n, launches, tot_ms, tot_fl, names = prof
        nm = n - 1                                   # MFMA kernel slots; the last slot is the HBM-bound GroupNorm pass
        # slots that ran the same kernel (e.g. 128- and 256-cout layers on one halo tile) are one kernel to rocprof too
        first = {}
        for i in range(nm):
            j = first.setdefault(names[i], i)
            if j != i:
                launches[j] += launches[i]; tot_ms[j] += tot_ms[i]; tot_fl[j] += tot_fl[i]
                launches[i] = 0; tot_ms[i] = 0.0; tot_fl[i] = 0.0
        dom = max(range(nm), key=lambda i: tot_ms[i])
        achieved = tot_fl[dom] / (tot_ms[dom] * 1e-3) / 1e12 if tot_ms[dom] > 0 else 0.0
        is8 = "fp8" in names[dom]
        dom_peak = MFMA_FP8_DENSE_PEAK_TFLOPS if is8 else MFMA_BF16_DENSE_PEAK_TFLOPS
        gemm_ms = sum(tot_ms[i] for i in range(nm))
        gn_gbs = tot_fl[nm] / (tot_ms[nm] * 1e-3) / 1e9 if tot_ms[nm] > 0 else 0.0
        traffic, traffic_src = pmc_traffic(names[dom], *traffic_key) if traffic_key else (None, None)
        fl8 = sum(tot_fl[i] for i in range(nm) if "fp8" in names[i])          # FLOPs that ran on the fp8 MFMA in this region
        roof = {"bound": "mfma", "kernel": names[dom], "achieved": round(achieved, 2), "peak": dom_peak, "unit": "TFLOP/s",
                "frac": round(achieved / dom_peak, 4),
                "peak_measured_bare_mfma_loop": MFMA_FP8_MEASURED_BARE_TFLOPS if is8 else MFMA_BF16_MEASURED_BARE_TFLOPS,
                "traffic": None if traffic is None else round(traffic), "traffic_unit": "bytes per launch (mean)",
                "traffic_source": traffic_src, "launches": int(launches[dom]),
                "avg_launch_ms": round(tot_ms[dom] / max(1, launches[dom]), 4),
                "all_mfma_kernels_share_of_step": round(gemm_ms / (elapsed * 1e3), 4),
                "per_config": {names[i]: {"launches": int(launches[i]), "ms": round(tot_ms[i], 3),
                                          "tflops": round(tot_fl[i] / max(tot_ms[i], 1e-9) / 1e9, 2)}
                               for i in range(nm) if launches[i]}}
        hbm = {"kernel": names[nm], "bound": "hbm", "achieved": round(gn_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(gn_gbs / HBM_PEAK_GBS, 4), "launches": int(launches[nm]),
               "avg_launch_ms": round(tot_ms[nm] / max(1, launches[nm]), 4),
               "share_of_step": round(tot_ms[nm] / (elapsed * 1e3), 4)}
        return roof, hbm, fl8

    def end_to_end(ips_per_gpu, flops_img, fl8_per_image):
        """End-to-end FLOP/s against the MFMA peak of the operands the FLOPs actually ran on: bf16 mode = the 2.5 PF dense bf16
        peak; fp8 mode = a FLOP-weighted mixed peak (time at peak = fp8 FLOPs / 5 PF + the rest / 2.5 PF)."""
        tf = ips_per_gpu * flops_img / 1e12
        ideal_s = (fl8_per_image / (MFMA_FP8_DENSE_PEAK_TFLOPS * 1e12) + (flops_img - fl8_per_image) / (MFMA_BF16_DENSE_PEAK_TFLOPS * 1e12))
        basis = ("2.5 PF dense bf16" if fl8_per_image == 0 else
                 f"FLOP-weighted mixed peak: {fl8_per_image / flops_img:.3f} of the FLOPs on the fp8 MFMA at 5 PF dense, the rest at 2.5 PF dense bf16")
        return {"end_to_end_tflops_per_gpu": round(tf, 2), "end_to_end_frac_of_mfma_peak": round(ips_per_gpu * ideal_s, 4),
                "end_to_end_peak_basis": basis}

    # ---- the headline region --------------------------------------------------------------------------
    if a.bucketed:
        plan = make_bucket_plan(a.warmup + a.steps)
        images_per_step = 2 * world * a.bucket_batch
        flops_step = [sum(pipe.flops_per_image(h, w) * n for (w, h, n) in b) for b, _ in plan[a.warmup:]]
        warm = [lambda p=p: bucket_step(*p) for p in plan[:a.warmup]]
        steps = [lambda p=p: bucket_step(*p) for p in plan[a.warmup:]]
    else:
        images_per_step = world * B
        flops_step = None
        warm = [plain_step] * a.warmup
        steps = [plain_step] * a.steps
    for f in warm:
        f()
    elapsed, out, prof = timed(steps)
    assert torch.isfinite(out).all()
    # the same K steps without the per-launch event records (the production path): reported beside the contract number
    with power:
        elapsed_plain, out, _ = timed(steps, profile=False)
    power_main = power.summary() if rank == 0 else None
    status = prof_ctx.status()
    assert status == 0, (f"vt_status = {status}: " + ("non-finite activations inside the encoder" if status & 1 else
                                                       "activations clamped to the e4m3 range (fp8 mode unsuitable for these weights)"))

    res = None
    if rank == 0:
        flops_img = pipe.flops_per_image(a.height, a.width)
        ips = images_per_step * a.steps / elapsed
        if flops_step is not None:
            flops_img = sum(flops_step) / (images_per_step * a.steps)      # mean over the drawn buckets
        roof, hbm, fl8 = roofline_of(prof, elapsed, None if (a.bucketed or a.encode_only) else (B, a.height, a.width, a.fp8))
        cfg = {"workload": (f"configs[3]: bucketed 512->1024 step 64 (67 reachable buckets), {2 * world} same-shape batches of "
                            f"{a.bucket_batch} per step, FLUX-VAE encode + 8-head attention decoder, {a.tags} tags"
                            if a.bucketed else
                            ("configs[4] (per GPU): " if a.fp8 else "configs[2]: " if not a.encode_only else "configs[1]: ")
                            + f"batch {B}/GPU {a.width}x{a.height} FLUX-VAE encode"
                            + ("" if a.encode_only else f" + 8-head attention decoder, {a.tags} tags"))
               + ", random-init weights (seeded), fp32 NCHW input resident in HBM",
               "global_batch": images_per_step, "parallelism": f"dp{world}", "tflop_per_image": round(flops_img / 1e12, 4)}
        cfg.update(end_to_end(ips / world, flops_img, fl8 / (images_per_step / world * a.steps)))
        res = {
            "metric": ("images/sec encode+tag, bucketed 512..1024 bf16" if a.bucketed else
                       "images/sec encode+tag, 1024^2 bf16" if not a.encode_only else "images/sec encode only, 1024^2 bf16").replace(
                           "bf16", "fp8 (3x3 convs, attention GEMMs, q|k / v projections; to_out, conv_in / conv_out bf16)" if a.fp8 else "bf16").replace(
                           "1024^2", "1024^2" if (a.height, a.width) == (1024, 1024) else f"{a.width}x{a.height}").replace(
                           "bf16", "fp16 conv operands (attention bf16)" if a.f16 else "bf16"),
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "ms_per_step_without_events": round(elapsed_plain / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8" if a.fp8 else ("f16" if a.f16 else "bf16"), "data": "synthetic",
            "config": cfg, "roofline": roof, "hbm_pass": hbm, "power": power_main,
            # the figure of merit under the 1400-W cap: socket power of the untraced steps / their images per second
            "joules_per_image": None if not power_main else round(power_main["socket_w_median"] / (images_per_step / world * a.steps / elapsed_plain), 3),
        }

    # ---- the other single-GPU configs of BASELINE.json on the same pipeline, attached to the ONE line as "also" (default run only):
    # configs[4] per GPU (fp8 mode, same batch) and configs[3] (bucketed batches); ~2 s of GPU time, headline keys untouched
    default_run = (world == 1 and not (a.fp8 or a.f16 or a.bucketed or a.encode_only or a.generic_conv or a.no_occ2 or a.flag)
                   and not a.no_also)
    also_f8_logits = also_f8_idx = also_f16 = None
    if default_run:
        also = {}
        K2, W2 = 10, 2
        pipe.ctx.call("vt_set_flag", 11, 1)
        try:
            for _ in range(W2):
                plain_step()
            with power:
                e8, out8, prof8 = timed([plain_step] * K2)
            power8 = power.summary()
            also_f8_logits = pipe.logits(x[:1])
            same8 = bool(torch.equal(also_f8_logits[0], out8[0]))
            st8 = prof_ctx.status()
        finally:
            pipe.ctx.call("vt_set_flag", 11, 0)
        roof8, hbm8, fl8_8 = roofline_of(prof8, e8, (B, a.height, a.width, True))
        ips8 = B * K2 / e8
        fimg = pipe.flops_per_image(a.height, a.width)
        also["configs4_fp8_per_gpu"] = {
            "workload": f"configs[4] (per GPU): batch {B}/GPU {a.width}x{a.height} encode+tag, {a.tags} tags, 3x3 convs, q|k / v projections and Q.K^T / P.V on e4m3 operands (vt_set_flag 11)",
            "value": round(ips8, 3), "unit": "images/sec", "steps": K2, "warmup": W2, "ms_per_step": round(e8 / K2 * 1e3, 3), "dtype": "fp8",
            "vt_status": st8, "identical_to_the_batched_result": same8,
            "roofline": {k: roof8[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "launches", "avg_launch_ms", "traffic", "traffic_source", "per_config")},
            "hbm_pass": {k: hbm8[k] for k in ("achieved", "frac", "share_of_step")},
            "power": None if power8 is None else {k: power8[k] for k in ("socket_w_median", "cap_w", "smu_sclk_mhz_median", "samples")}}
        also["configs4_fp8_per_gpu"].update(end_to_end(ips8, fimg, fl8_8 / (B * K2)))
        also["configs4_fp8_per_gpu"]["joules_per_image"] = None if power8 is None else round(power8["socket_w_median"] / ips8, 3)
        also_f8_idx = pipe.confidence(also_f8_logits)[1]
        # configs[1]: batch 8, 1024^2, encoder only (vae_model.encode on the VAE mirror's own context), bf16
        x1 = x[:8] if B >= 8 else x
        enc_step = lambda: vae_model.encode(x1)
        for _ in range(W2):
            enc_step()
        prof_main, prof_ctx = prof_ctx, vae._context()
        try:
            with power:
                e1, out1, prof1 = timed([enc_step] * K2)
            power1 = power.summary()
            st1 = prof_ctx.status()
        finally:
            prof_ctx = prof_main
        assert torch.isfinite(out1).all() and st1 == 0
        roof1, hbm1, _ = roofline_of(prof1, e1)
        ips1 = x1.shape[0] * K2 / e1
        also["configs1_encode_only"] = {
            "workload": f"configs[1]: batch {x1.shape[0]} {a.width}x{a.height} FLUX-VAE encode only (DiffusersVAEWrapper.encode), bf16",
            "value": round(ips1, 3), "unit": "images/sec", "steps": K2, "warmup": W2, "ms_per_step": round(e1 / K2 * 1e3, 3), "dtype": "bf16",
            "roofline": {k: roof1[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "launches", "avg_launch_ms")},
            "hbm_pass": {k: hbm1[k] for k in ("achieved", "frac", "share_of_step")},
            "joules_per_image": None if power1 is None else round(power1["socket_w_median"] / ips1, 3)}
        also["configs1_encode_only"].update(end_to_end(ips1, fimg, 0.0))
        plan3 = make_bucket_plan(W2 + K2)
        for p3 in plan3[:W2]:
            bucket_step(*p3)
        e3, out3, prof3 = timed([lambda p=p: bucket_step(*p) for p in plan3[W2:]])
        assert torch.isfinite(out3).all()
        roof3, hbm3, _ = roofline_of(prof3, e3)
        n3 = 2 * a.bucket_batch * K2
        f3 = sum(pipe.flops_per_image(h, w) * n for b3, _ in plan3[W2:] for (w, h, n) in b3) / n3
        also["configs3_bucketed"] = {
            "workload": f"configs[3] on one GPU: bucketed 512->1024 step 64, 2 same-shape batches of {a.bucket_batch} per step, {a.tags} tags, bf16",
            "value": round(n3 / e3, 3), "unit": "images/sec", "steps": K2, "warmup": W2, "ms_per_step": round(e3 / K2 * 1e3, 3), "dtype": "bf16",
            "tflop_per_image_mean": round(f3 / 1e12, 4),
            "roofline": {k: roof3[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "launches", "avg_launch_ms")},
            "hbm_pass": {k: hbm3[k] for k in ("achieved", "frac", "share_of_step")}}
        also["configs3_bucketed"].update(end_to_end(n3 / e3, f3, 0.0))
        bucket_cache.clear()
        # the precision mode (vt_set_flag 18: fp16 instead of bf16 MFMA operands for the convolutions) on the headline batch: what it costs, and
        # -- filled in by the parity block below -- what it buys on a SMOOTH picture, where bf16 operands leave north_star's 1e-2
        pipe.set_fp16_operands(True)
        try:
            for _ in range(W2):
                plain_step()
            with power:
                e16, out16, prof16 = timed([plain_step] * K2)
            power16 = power.summary()
            also_f16 = pipe.logits(x[:1], return_latent=True)
            st16 = prof_ctx.status()
        finally:
            pipe.set_fp16_operands(False)
        assert torch.isfinite(out16).all() and st16 == 0
        roof16, hbm16, _ = roofline_of(prof16, e16)
        ips16 = B * K2 / e16
        also["precision_mode_fp16_operands"] = {
            "workload": f"batch {B}/GPU {a.width}x{a.height} encode+tag, {a.tags} tags, fp16 instead of bf16 MFMA operands for every convolution (vt_set_flag 18; attention bf16)",
            "value": round(ips16, 3), "unit": "images/sec", "steps": K2, "warmup": W2, "ms_per_step": round(e16 / K2 * 1e3, 3), "dtype": "f16",
            "roofline": {k: roof16[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "launches", "avg_launch_ms")},
            "joules_per_image": None if power16 is None else round(power16["socket_w_median"] / ips16, 3)}
        also["precision_mode_fp16_operands"].update(end_to_end(ips16, fimg, 0.0))
        res["also"] = also

    if rank == 0:
        if world == 1 and not a.no_cpu_baseline and not a.bucketed:
            # parity half of the metric: image 0 of the measured batch through the HIP path (outside the timed region) against
            # the CPU oracle's warm-up run on the same image
            xs = x[:1]
            if a.encode_only:
                lat_g, lg_g = vae_model.encode(xs), None
            else:
                lg_g, lat_g = pipe.logits(xs, return_latent=True)
            same = torch.equal(lg_g[0], out[0]) if lg_g is not None else torch.equal(lat_g[0], out[0])
            (ref_lat, ref_lg), res["cpu_baseline"] = cpu_baseline(xs.cpu(), a.tags, flops_img)
            dlat = float(f"{(lat_g.cpu() - ref_lat).abs().max().item():.3e}")
            res["max_abs_dlatent"] = dlat
            tol = 1e-2
            par = {"vs": "oracle/ (CPU fp32 restatement; encoder half unpinned against diffusers, see DESIGN.md section 2)",
                   "image": "image 0 of the measured batch", "tolerance": tol, "identical_to_the_batched_result": bool(same),
                   "weights": "PyTorch-init synthetic (parity on a trained checkpoint is unpinned: none is available offline)"}
            if lg_g is not None:
                dlg = float(f"{(lg_g.cpu() - ref_lg).abs().max().item():.3e}")
                res["max_abs_dlogit"] = dlg
                par["logits_within_tolerance"] = bool(dlg <= tol)
                # the metric's second half: the HIP path's sorted tag indices (vt_get_confidence on its own logits) against the oracle's order
                from oracle.agreement import argsort_agreement
                par["argsort_agreement"] = argsort_agreement(ref_lg[0], lg_g[0], pipe.confidence(lg_g)[1][0])
            if a.fp8:
                par["latents_within_tolerance"] = None
                par["latents_note"] = "fp8 mode claims the logits only (north_star's fp8 line); its latents (~1e-1) are out of scope, infer_vae stays bf16"
            else:
                par["latents_within_tolerance"] = bool(dlat <= tol)
            res["parity"] = par
            if also_f16 is not None:
                m16 = res["also"]["precision_mode_fp16_operands"]
                m16["max_abs_dlogit"] = float(f"{(also_f16[0].cpu() - ref_lg).abs().max().item():.3e}")
                m16["max_abs_dlatent"] = float(f"{(also_f16[1].cpu() - ref_lat).abs().max().item():.3e}")
                # a smooth picture (a random 12 x 12 field upsampled bicubically + a little noise, clamped, 8 bits per channel like a decoded file),
                # 512^2: the HIP path with bf16 and with fp16 operands against the fp32 oracle; the oracle with bf16-rounded operands beside them
                import torch.nn.functional as F
                from oracle import decoder_ref, encoder_ref
                g = torch.Generator().manual_seed(7)
                pic = F.interpolate(torch.rand(1, 3, 12, 12, generator=g), size=(512, 512), mode="bicubic", align_corners=False)
                pic = ((pic + 0.06 * torch.randn(1, 3, 512, 512, generator=g)).clamp(0, 1) * 255).to(torch.uint8)
                pic = (pic.float() / 255.0 - 0.5) / 0.5
                sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
                with torch.no_grad():
                    lat_ref = encoder_ref.vae_wrapper_encode(sd_e, pic)
                    lat_emu = encoder_ref.vae_wrapper_encode(sd_e, pic, emulate_bf16=True)
                lat_b = pipe.logits(pic.to(dev), return_latent=True)[1].cpu()
                pipe.set_fp16_operands(True)
                try:
                    lat_h = pipe.logits(pic.to(dev), return_latent=True)[1].cpu()
                finally:
                    pipe.set_fp16_operands(False)
                e3 = lambda t: float(f"{t:.3e}")
                res["parity"]["smooth_picture_512"] = {
                    "what": "max |dlatent| against the fp32 oracle on a smooth 512x512 picture (flat regions make neighbouring pixels' rounding errors equal: "
                            "a 3x3 conv adds them coherently) -- bf16 operands leave north_star's 1e-2 there, in this path and in the oracle's own bf16 emulation",
                    "hip_bf16_operands": e3((lat_b - lat_ref).abs().max().item()), "hip_fp16_operands": e3((lat_h - lat_ref).abs().max().item()),
                    "oracle_with_bf16_rounded_operands": e3((lat_emu - lat_ref).abs().max().item()),
                    "rms_hip_bf16_operands": e3((lat_b - lat_ref).pow(2).mean().sqrt().item()),
                    "fp16_operands_within_tolerance": bool((lat_h - lat_ref).abs().max().item() <= tol)}
            if also_f8_logits is not None:
                d8 = float(f"{(also_f8_logits.cpu() - ref_lg).abs().max().item():.3e}")
                res["also"]["configs4_fp8_per_gpu"]["max_abs_dlogit"] = d8
                res["also"]["configs4_fp8_per_gpu"]["logits_within_tolerance"] = bool(d8 <= tol)
                res["also"]["configs4_fp8_per_gpu"]["argsort_agreement"] = argsort_agreement(ref_lg[0], also_f8_logits[0], also_f8_idx[0])
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
