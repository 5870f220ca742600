"""GPU parity of the whole hot path through the reference-shaped Python surface:
encoder vs the committed goldens / the CPU oracle, decoder vs outputs of the reference itself,
and size-independent properties at BASELINE.json's full sizes."""
import pytest
import torch

from oracle import decoder_ref, encoder_ref
from oracle.agreement import argsort_agreement
from vae_tagger_amd import synth

from _util import golden, latent_input

pytestmark = pytest.mark.gpu

TOL_LATENT_BF16 = 1e-2      # north_star: latent / logit tensors within 1e-2 for the bf16 path
TOL_LOGIT_F32 = 1e-3        # decoder runs in fp32: within 1e-3 (observed ~1e-5)
# fp8 mode (configs[4]) claims the LOGITS (1e-2); its latents are outside north_star's tolerance by design.  These two are regression
# bounds at what is observed (max 0.10-0.12, rms 0.021 over 256^2 .. 1024^2 and all 16 images of the bench batch), not a parity claim.
FP8_LATENT_MAX, FP8_LATENT_RMS = 0.13, 0.025
FP8_ATTN_TOL = 8e-3          # attention output (+ residual) on e4m3 P against fp32 attention of the same e4m3 q, k, v (observed below)


@pytest.fixture(scope="module")
def vae():
    from vae_tagger_amd.diffusers_vae_loader import (DiffusersVAEWrapper, get_diffusers_vae_config,
                                                      load_diffusers_vae_from_config)
    m = load_diffusers_vae_from_config(get_diffusers_vae_config())
    missing, unexpected = m.load_state_dict(synth.synth_state_dict(synth.encoder_manifest(), seed=0), strict=False)
    assert not missing and not unexpected
    return DiffusersVAEWrapper(m).to("cuda").eval()


def _check_tag_order(pipe, logits_row, ref_logits_row, what):
    """The metric's second half (north_star: "bit-exact tag-index argsort"; reference modules.py:470-475 consumed at
    infer_full.py:106-125): the HIP path's sorted tag indices for one image against the oracle's.  Asserted: identical indices at
    every rank whose oracle logit is more than 2 x the measured max |dlogit| from both neighbours, every displaced tag inside that
    band, the thresholded (0.5) tag set equal outside the band, and the device order = (logit desc, index asc) of the device
    logits exactly.  Printed: the fraction of ranks that could be compared and the first disagreeing rank."""
    conf, idx = pipe.confidence(logits_row)
    ag = argsort_agreement(ref_logits_row[0], logits_row[0], idx[0])
    print(f"{what}: argsort agreement: {ag['ranks_compared']}/{ag['ranks']} ranks comparable at 2 x max|dlogit| {ag['max_abs_dlogit']:.2e} "
          f"-> identical there: {ag['identical_at_compared_ranks']} (first disagreeing compared rank: {ag['first_disagreeing_compared_rank']}); "
          f"identical positions overall {ag['frac_identical_positions']:.4f}, first differing rank {ag['first_differing_rank']}, "
          f"max displacement {ag['max_rank_displacement']}, top-1 / top-5 / top-10 sets same: {ag['top1_identical']} / {ag['top5_set_identical']} / "
          f"{ag['top10_set_identical']}; tags >= 0.5: oracle {ag['tags_above_threshold_oracle']}, HIP {ag['tags_above_threshold_measured']}")
    assert ag["identical_at_compared_ranks"], ag
    assert ag["swaps_stay_inside_the_2d_band"] and ag["threshold_set_matches_outside_the_band"], ag
    assert torch.equal(idx.cpu(), decoder_ref.get_confidence(logits_row.cpu())[1])
    assert torch.allclose(conf.cpu(), torch.sigmoid(logits_row.cpu()).gather(1, idx.cpu()), atol=1e-6)
    return ag


def _decoder(n, flags=(True, True, False), plain=False):
    from vae_tagger_amd.modules import ClassificationDecoder, create_attention_decoder
    if plain:
        d = ClassificationDecoder(16, 16, 16, n)
        d.load_state_dict(synth.synth_state_dict(synth.plain_decoder_manifest(n), seed=1), strict=False)
    else:
        d = create_attention_decoder(16, 16, 16, n, {"use_spatial_attention": flags[0], "use_self_attention": flags[1],
                                                     "use_cross_attention": flags[2], "attention_heads": 8})
        missing, unexpected = d.load_state_dict(
            synth.synth_state_dict(synth.attention_decoder_manifest(n, 16, *flags), seed=1), strict=False)
        assert not missing and not unexpected
    return d.to("cuda").eval()


@pytest.mark.parametrize("name,h,w", [("enc_64x64", 64, 64), ("enc_128x192", 128, 192), ("enc_512x512", 512, 512)])
def test_encoder_matches_golden_latents(vae, name, h, w):
    g = golden("encoder_" + name)
    x = synth.synth_images(1, h, w, seed=3)
    lat = vae.encode(x.cuda()).cpu()
    assert lat.shape == g["latent"].shape
    err = (lat - g["latent"]).abs().max().item()
    assert err <= TOL_LATENT_BF16, f"max |dlatent| = {err}"


@pytest.mark.parametrize("h,w", [(72, 88), (100, 76)])
def test_encoder_matches_oracle_on_odd_shapes(vae, h, w):
    """non-multiple-of-64 (and non-multiple-of-8) inputs: ragged tiles, odd downsample sizes, S % 8 != 0."""
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    x = synth.synth_images(2, h, w, seed=11)
    ref = encoder_ref.vae_wrapper_encode(sd, x)
    lat = vae.encode(x.cuda()).cpu()
    assert lat.shape == ref.shape == (2, 16, h // 8, w // 8)
    assert (lat - ref).abs().max().item() <= TOL_LATENT_BF16


@pytest.mark.parametrize("flags", [(0, 0, 0, 1, 1, 1), (1, 0, 0, 1, 1, 0), (1, 1, 0, 1, 1, 1), (1, 1, 1, 1, 0, 1), (1, 1, 0, 0, 0, 0), (1, 1, 0, 1, 0, 1),
                                   (1, 1, 0, 2, 1, 1), (1, 1, 0, 3, 1, 1), (1, 0, 0, 3, 0, 0), (1, 1, 0, 3, 1, 1, 0, 1), (1, 1, 0, 3, 1, 1, 1, 2),
                                   (1, 1, 0, 3, 1, 1, 1, 0, 0), (1, 1, 0, 0, 1, 1, 1, 0, 1), (1, 1, 0, 1, 0, 1, 1, 0, 1)])
def test_encoder_kernel_variants_agree(vae, flags):
    """flags: (halo conv kernel, epilogue GroupNorm statistics, fused apply, 2-workgroup tile mode 0..3, fp16 residual storage, MFMA conv_in
    [, short-K GEMM tile, attention softmax mode 0..2, conv_shortcut fused into conv2]):
    every combination stays within the bf16 tolerance of the fp32 oracle."""
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    x = synth.synth_images(2, 96, 160, seed=17)
    ref = encoder_ref.vae_wrapper_encode(sd, x)
    ctx = vae.vae._context()
    try:
        for f, v in enumerate(flags):
            ctx.call("vt_set_flag", f, v)
        lat = vae.encode(x.cuda()).cpu()
    finally:
        for f, v in enumerate((1, 1, 0, 3, 1, 1, 1, 0, 1)):
            ctx.call("vt_set_flag", f, v)
    assert (lat - ref).abs().max().item() <= TOL_LATENT_BF16


def test_autoencoderkl_surface_moments_mode_sample(vae):
    x = synth.synth_images(2, 64, 64, seed=5).cuda()
    post = vae.vae.encode(x).latent_dist
    assert post.parameters.shape == (2, 32, 8, 8)
    scaled = vae.encode(x)
    assert torch.allclose(post.mode() * 0.3611 + 0.1159, scaled, atol=1e-6)
    assert post.sample().shape == (2, 16, 8, 8) and post.kl().shape == (2,)
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    ref = encoder_ref.encoder_moments(sd, x.cpu())
    assert (post.parameters.cpu() - ref).abs().max().item() <= 3e-2     # un-scaled moments: 1e-2 / 0.3611


DEC = [("attn_n11_16x16", 11, (2, 16, 16, 16), (True, True, False)),
       ("attn_n10000_64x64", 10000, (2, 16, 64, 64), (True, True, False)),
       ("attn_n11_72x128", 11, (1, 16, 72, 128), (True, True, False)),
       ("attn_cross_n11_16x16", 11, (2, 16, 16, 16), (True, True, True)),
       ("attn_nospatial_n11_16x16", 11, (2, 16, 16, 16), (False, True, False))]


@pytest.mark.parametrize("name,n,shape,flags", DEC)
def test_decoder_matches_reference_outputs(name, n, shape, flags):
    g = golden("decoder_" + name)
    dec = _decoder(n, flags)
    x = latent_input(shape, seed=7).cuda()
    logits = dec(x).cpu()
    err = (logits - g["logits"]).abs().max().item()
    assert err <= TOL_LOGIT_F32, f"max |dlogit| = {err}"
    conf, idx = dec.get_confidence(x)
    conf, idx = conf.cpu(), idx.cpu()
    assert torch.allclose(conf, g["conf_sorted"], atol=1e-5)
    # bit-exact tag-index argsort wherever adjacent confidences differ by more than the fp32 tolerance
    gap = (g["conf_sorted"][:, :-1] - g["conf_sorted"][:, 1:]).abs()
    distinct = torch.ones_like(idx, dtype=torch.bool)
    distinct[:, :-1] &= gap > 1e-5
    distinct[:, 1:] &= gap > 1e-5
    assert distinct.float().mean() > 0.5
    assert torch.equal(idx[distinct], g["indices"][distinct])
    # the device sort is exactly (logit desc, index asc) of the device logits
    ref_conf, ref_idx = decoder_ref.get_confidence(logits)
    assert torch.equal(idx, ref_idx)


def test_plain_decoder_matches_reference_outputs():
    g = golden("decoder_plain_n11_16x16")
    dec = _decoder(11, plain=True)
    logits = dec(latent_input((2, 16, 16, 16), seed=7).cuda()).cpu()
    assert (logits - g["logits"]).abs().max().item() <= TOL_LOGIT_F32


def test_sort_edge_cases():
    dec = _decoder(11)
    for n in (1, 2, 3, 1000, 16384, 16385, 20000, 40000, 70001):     # > 16384: per-block LDS sorts + global merge passes
        lg = torch.randn(3, n, generator=torch.Generator().manual_seed(n))
        lg[:, : n // 2] = lg[:, : n // 2].round()             # many exact ties
        lg[0, 0] = 30.0                                       # sigmoid saturates to 1.0 here
        conf, idx = dec.confidence_from_logits(lg.cuda())
        rc, ri = decoder_ref.get_confidence(lg)
        assert torch.equal(idx.cpu(), ri) and torch.allclose(conf.cpu(), rc, atol=1e-6)


def test_sort_places_nan_logits_last():
    """NaN logits (a broken checkpoint) sort after every real value, -inf included, with their own tag index -- never a
    padding entry -- and the row still holds every tag exactly once."""
    dec = _decoder(11)
    for n in (10, 10000, 20000):
        lg = torch.randn(2, n, generator=torch.Generator().manual_seed(n))
        lg[0, 3] = float("nan"); lg[0, 7] = float("nan"); lg[0, 5] = float("-inf"); lg[1, 0] = float("inf")
        conf, idx = dec.confidence_from_logits(lg.cuda())
        conf, idx = conf.cpu(), idx.cpu()
        assert sorted(idx[0].tolist()) == list(range(n)) and sorted(idx[1].tolist()) == list(range(n))
        assert idx[0, -2:].tolist() == [3, 7] and torch.isnan(conf[0, -2:]).all() and idx[0, -3] == 5 and conf[0, -3] == 0.0
        assert idx[1, 0] == 0 and conf[1, 0] == 1.0 and torch.isfinite(conf[1]).all()
        real = conf[0, :-2]
        assert (real[:-1] >= real[1:]).all()


def test_device_summary_matches_the_host_loop(vae):
    """vt_summarize_confidence (threshold count, top-k, max, top-5 mean on the device; infer_full.py:106-125) against the
    reference formulation run on a full host copy: identical JSON entries, including more tags above the threshold than
    the summary carries and fewer than five tags."""
    from vae_tagger_amd import infer_full
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    pipe = EncodeTagPipeline(vae, _decoder(11))
    for n, thr in ((3, 0.5), (40, 0.5), (40, 0.0), (10000, 0.5), (10000, 0.9), (10000, 0.99999), (20000, 0.3)):
        lg = 2.0 * torch.randn(3, n, generator=torch.Generator().manual_seed(n))
        lg[1] -= 6.0                                          # an image with (almost) nothing above the threshold
        conf, idx = pipe.confidence(lg.cuda())
        names = [f"t{i}" for i in range(n)]
        got = infer_full.summarize_batch(pipe, conf, idx, names, thr)
        ch, ih = conf.cpu().numpy(), idx.cpu().numpy()
        want = [infer_full.summarize(ch[b], ih[b], names, thr) for b in range(3)]
        assert got == want, (n, thr)
    lg = torch.randn(2, 50)
    lg[1, 4] = float("nan")
    conf, idx = pipe.confidence(lg.cuda())
    with pytest.raises(FloatingPointError):
        infer_full.summarize_batch(pipe, conf, idx, [f"t{i}" for i in range(50)], 0.5)


def test_encode_tag_pipeline_matches_oracle(vae):
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    dec = _decoder(1000)
    pipe = EncodeTagPipeline(vae, dec)
    x = synth.synth_images(3, 128, 192, seed=21)
    logits, lat = pipe.logits(x.cuda(), return_latent=True)
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(1000), seed=1)
    ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x)
    ref_logits = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
    assert (lat.cpu() - ref_lat).abs().max().item() <= TOL_LATENT_BF16
    assert (logits.cpu() - ref_logits).abs().max().item() <= 1e-2       # north_star: logits within 1e-2
    for i in range(3):
        _check_tag_order(pipe, logits[i:i + 1], ref_logits[i:i + 1], f"128x192 image {i}")
    # same result through the two separate objects (reference call order, infer_full.py:101-102)
    lat2 = vae.encode(x.cuda())
    assert torch.equal(lat2, lat)
    assert torch.equal(dec(lat2), logits)


def test_full_size_properties(vae):
    """BASELINE.json configs[1] size (1024^2): size-independent properties -- determinism, batch-permutation equivariance,
    batch-composition invariance (the oracle comparison at this size is test_config2_batch16_1024_...)."""
    x = synth.synth_images(3, 1024, 1024, seed=31).cuda()
    a = vae.encode(x)
    b = vae.encode(x)
    assert torch.equal(a, b), "non-deterministic"
    assert torch.isfinite(a).all() and a.shape == (3, 16, 128, 128)
    perm = torch.tensor([2, 0, 1], device="cuda")
    assert torch.equal(vae.encode(x[perm]), a[perm])
    assert torch.equal(vae.encode(x[1:2]), a[1:2])
    # down-scaled copy of a 512^2 golden region is NOT expected to match; instead tie the big shape to the
    # oracle through statistics the goldens recorded at 512^2 (same weights, same input distribution)
    g = golden("encoder_enc_512x512")["latent"]
    assert abs(a.mean().item() - g.mean().item()) < 0.02 and abs(a.std().item() - g.std().item()) < 0.02


def test_config2_batch16_1024_matches_oracle_across_attention_groups(vae):
    """BASELINE.json configs[2]: batch 16 x 1024^2, 10 000 tags.  At this size the mid-block attention runs as two launch
    groups of 8 images: image 0 (first group) and image 11 (second group) are compared with the CPU oracle at the north_star
    tolerance, and the result of an image must not depend on which group / batch position it ran in (bit-exact)."""
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    n = 10000
    dec = _decoder(n)
    pipe = EncodeTagPipeline(vae, dec)
    x = synth.synth_images(16, 1024, 1024, seed=77)
    xd = x.cuda()
    logits, lat = pipe.logits(xd, return_latent=True)
    assert torch.isfinite(logits).all() and pipe.status() == 0
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n), seed=1)
    for i in (0, 11):
        ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x[i:i + 1])
        ref_logits = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
        dl = (lat[i:i + 1].cpu() - ref_lat).abs().max().item()
        dg = (logits[i:i + 1].cpu() - ref_logits).abs().max().item()
        print(f"configs[2] image {i}: max|dlatent| {dl:.3e} max|dlogit| {dg:.3e}")
        assert dl <= TOL_LATENT_BF16 and dg <= 1e-2, (i, dl, dg)
        ag = _check_tag_order(pipe, logits[i:i + 1], ref_logits, f"configs[2] image {i}")
        assert ag["top1_identical"] or ag["max_abs_dlogit"] > 0           # (top-1 is reported; within the band it may legitimately differ)
    # group-boundary / batch-composition invariance, bit for bit
    assert torch.equal(vae.encode(xd[8:9]), lat[8:9])
    assert torch.equal(vae.encode(xd[2:4]), lat[2:4])             # 1 / 2 images: Q.K^T splits each query block's key sweep over 4 / 2 workgroups
    assert torch.equal(vae.encode(xd[11:12]), lat[11:12])
    l2, lat2 = pipe.logits(xd[4:13], return_latent=True)     # 9 images: groups of 5 + 4, image 8 now sits in the first group
    assert torch.equal(lat2, lat[4:13]) and torch.equal(l2, logits[4:13])


# BASELINE.json configs[3]: same-shape batches from the reference's 512..1024 step-64 aspect buckets (modules.py:180-222);
# (width, height): the largest reachable bucket, the two most oblong ones, and two mid-sized ones
BUCKETS = [(960, 1024), (512, 1024), (1024, 512), (832, 640), (576, 768)]


@pytest.mark.parametrize("w,h", BUCKETS)
def test_config3_bucket_shapes_match_oracle(vae, w, h):
    from vae_tagger_amd.modules import AspectRatioBucketing
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    bk = AspectRatioBucketing(512, 1024, 64)
    assert (w, h) in bk.buckets
    n = 1000
    pipe = EncodeTagPipeline(vae, _decoder(n))
    x = synth.synth_images(2, h, w, seed=w * 4096 + h)
    logits, lat = pipe.logits(x.cuda(), return_latent=True)
    assert lat.shape == (2, 16, h // 8, w // 8)
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n), seed=1)
    ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x[1:2])
    ref_logits = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
    dl = (lat[1:2].cpu() - ref_lat).abs().max().item()
    dg = (logits[1:2].cpu() - ref_logits).abs().max().item()
    print(f"bucket {w}x{h}: max|dlatent| {dl:.3e} max|dlogit| {dg:.3e}")
    assert dl <= TOL_LATENT_BF16 and dg <= 1e-2
    assert torch.equal(pipe.logits(x[1:2].cuda()), logits[1:2])


def test_fp16_overflow_trips_the_status_word_and_fp32_storage_matches_oracle():
    """The residual stream is STORED as fp16 by default.  With conv_in scaled until its outputs exceed +-65504 the stored
    stream holds inf: the GroupNorm finalize kernel raises the sticky status bit (vt_status) and the outputs are not finite;
    with fp32 storage (vt_set_flag(ctx, 4, 0)) the same weights stay on the CPU oracle."""
    from vae_tagger_amd.diffusers_vae_loader import (DiffusersVAEWrapper, get_diffusers_vae_config,
                                                      load_diffusers_vae_from_config)
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd["encoder.conv_in.weight"] = sd["encoder.conv_in.weight"] * 4.0e5
    sd["encoder.conv_in.bias"] = sd["encoder.conv_in.bias"] * 4.0e5
    m = load_diffusers_vae_from_config(get_diffusers_vae_config())
    m.load_state_dict(sd, strict=False)
    w = DiffusersVAEWrapper(m).to("cuda").eval()
    x = synth.synth_images(2, 64, 64, seed=9)
    ref = encoder_ref.vae_wrapper_encode(sd, x)
    assert encoder_ref.encoder_moments(sd, x, taps=(t := {})) is not None and t["conv_in"].abs().max() > 65504
    assert m.status() == 0
    with pytest.raises(FloatingPointError):                    # the reference-shaped wrapper checks by default: finite latents or an exception
        w.encode(x.cuda())
    assert m.status() == 0                                     # (the check read and cleared the word)
    w.check_finite = False                                     # callers that poll the word themselves
    lat = w.encode(x.cuda())
    assert m.status() == 1 and m.status() == 0                 # sticky until read, cleared by the read
    assert not torch.isfinite(lat).all()
    m.check_finite = True                                      # the same switch on the diffusers-shaped object
    with pytest.raises(FloatingPointError):
        w.encode(x.cuda())
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    pipe = EncodeTagPipeline(w, _decoder(11))                  # the fused pipeline checks by default too (round 4): no silent non-finite tags
    with pytest.raises(FloatingPointError):
        pipe.tag(x.cuda())
    assert pipe.status() == 0
    pipe.check_finite = False                                  # ... unless the caller polls the word itself (the CLIs, bench.py)
    _, lat_p = pipe.logits(x.cuda(), return_latent=True)       # (the TAGS of non-finite latents can look perfectly finite: only the word tells)
    assert pipe.status() == 1 and pipe.status() == 0 and not torch.isfinite(lat_p).all()
    m.set_fp32_residual(True)
    lat32 = w.encode(x.cuda())                                 # check_finite still on: no raise
    assert m.status() == 0
    assert (lat32.cpu() - ref).abs().max().item() <= TOL_LATENT_BF16


def test_fp8_saturation_trips_its_own_status_bit():
    """fp8 mode clamps to the e4m3 range (+-448 after the activation scale of 8) silently.  With one GroupNorm's affine scaled by 100
    the normalised activations reach ~+-400: in fp8 mode bit 1 of vt_status (VT_STATUS_FP8_SATURATED) is raised -- sticky, cleared by
    the read -- and nothing else; on the bf16 path the same weights leave the word at zero."""
    from vae_tagger_amd._lib import VT_STATUS_FP8_SATURATED
    from vae_tagger_amd.diffusers_vae_loader import (DiffusersVAEWrapper, get_diffusers_vae_config,
                                                      load_diffusers_vae_from_config)
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    for k in ("weight", "bias"):
        sd[f"encoder.down_blocks.0.resnets.0.norm1.{k}"] = sd[f"encoder.down_blocks.0.resnets.0.norm1.{k}"] * 100.0
    m = load_diffusers_vae_from_config(get_diffusers_vae_config())
    m.load_state_dict(sd, strict=False)
    w = DiffusersVAEWrapper(m).to("cuda").eval()
    w.check_finite = False                                     # this test reads the word itself
    x = synth.synth_images(2, 128, 96, seed=5).cuda()
    ctx = m._context()
    lat = w.encode(x)
    assert m.status() == 0 and torch.isfinite(lat).all()
    try:
        ctx.call("vt_set_flag", 11, 1)
        lat8 = w.encode(x)
        assert m.status() == VT_STATUS_FP8_SATURATED and m.status() == 0
        assert torch.isfinite(lat8).all()
    finally:
        ctx.call("vt_set_flag", 11, 0)


@pytest.mark.parametrize("res", [256, 512, 1024])
def test_config4_fp8_operands_keep_the_logits_within_tolerance(vae, res):
    """BASELINE.json configs[4] (opt-in, vt_set_flag(ctx, 11, 1)): the 20 stride-1 3x3 resnet convs on fp8 e4m3 operands.
    north_star's fp8 target line constrains the LOGITS (within 1e-2 of the CPU reference); the latents are documented to move by
    up to ~1e-1 (tests/diagnostics/fp8_study.py: max 0.087, rms 0.019 at 512^2 for the same rounding emulated on the CPU)."""
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    n = 10000
    pipe = EncodeTagPipeline(vae, _decoder(n))
    x = synth.synth_images(2, res, res, seed=3)
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n), seed=1)
    nref = 1 if res >= 1024 else 2                            # (the CPU oracle needs ~10 s per 1024^2 image: one of the two there)
    ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x[:nref])
    ref_logits = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
    try:
        pipe.ctx.call("vt_set_flag", 11, 1)
        logits, lat = pipe.logits(x.cuda(), return_latent=True)
        again = pipe.logits(x.cuda())
    finally:
        pipe.ctx.call("vt_set_flag", 11, 0)
    assert pipe.status() == 0 and torch.equal(again, logits)
    dl = (lat[:nref].cpu() - ref_lat)
    dg = (logits[:nref].cpu() - ref_logits).abs().max().item()
    print(f"fp8 {res}^2: max|dlatent| {dl.abs().max():.3e} rms {dl.pow(2).mean().sqrt():.3e}  max|dlogit| {dg:.3e}")
    assert dg <= 1e-2
    assert dl.abs().max().item() <= FP8_LATENT_MAX and dl.pow(2).mean().sqrt().item() <= FP8_LATENT_RMS
    bf16_logits = pipe.logits(x.cuda())                       # back on bf16 operands: the tight tolerance again
    assert (bf16_logits[:nref].cpu() - ref_logits).abs().max().item() <= 1e-3


@pytest.mark.parametrize("shape", [1, 2, 5, 6])
def test_config4_fp8_tile_shapes_through_the_encoder(vae, shape):
    """vt_set_flag(ctx, 16, shape): the 8-wave tiles of the fp8 halo conv inside the whole path (GroupNorm partials per tile, fp16 residual
    staging of eight waves, the fused shortcut on layers with Cin > 128 for shape | 4), ragged 100 x 148 and whole-tile 128 x 192 inputs:
    logits within 1e-2 of the oracle and deterministic.  The conv outputs are bit-identical between shapes (test_fp8_conv_tile_shapes_are_bit_identical);
    the GroupNorm partials are per tile, so the last bit of a (scale, shift) may differ -- in fp8 mode that flips e4m3 roundings, an effect
    of the size of the mode's own error (tests/diagnostics/fp8_tile_shape_diff.py: latents equal bit for bit or up to 6e-2 apart)."""
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    n = 1000
    pipe = EncodeTagPipeline(vae, _decoder(n))
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n), seed=1)
    for (hh, ww) in ((100, 148), (128, 192)):
        x = synth.synth_images(2, hh, ww, seed=hh + shape)
        ref_logits = decoder_ref.attention_decoder_forward(sd_d, encoder_ref.vae_wrapper_encode(sd_e, x))
        try:
            pipe.set_fp8(True)
            base = pipe.logits(x.cuda())
            pipe.ctx.call("vt_set_flag", 16, shape)
            logits = pipe.logits(x.cuda())
            again = pipe.logits(x.cuda())
        finally:
            pipe.ctx.call("vt_set_flag", 16, 0)
            pipe.set_fp8(False)
        assert pipe.status() == 0 and torch.equal(again, logits)
        assert (logits.cpu() - ref_logits).abs().max().item() <= 1e-2
        assert (logits - base).abs().max().item() <= 1e-2


@pytest.mark.parametrize("h,w,b", [(72, 88, 2), (100, 76, 1), (576, 768, 2), (1024, 512, 1)])
def test_config4_fp8_on_ragged_and_bucket_shapes(vae, h, w, b):
    """fp8 mode on shapes that are not multiples of its 8 x 32 pixel tile (ragged tiles in every stage, odd sizes into the
    stride-2 convs) and on two aspect-ratio buckets: logits within 1e-2 of the oracle, and the mode is deterministic."""
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    n = 1000
    pipe = EncodeTagPipeline(vae, _decoder(n))
    x = synth.synth_images(b, h, w, seed=h * 7 + w)
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n), seed=1)
    ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x[:1])
    ref_logits = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
    try:
        pipe.ctx.call("vt_set_flag", 11, 1)
        logits, lat = pipe.logits(x.cuda(), return_latent=True)
        again = pipe.logits(x.cuda())
    finally:
        pipe.ctx.call("vt_set_flag", 11, 0)
    assert pipe.status() == 0 and torch.equal(again, logits)
    assert lat.shape == (b, 16, h // 8, w // 8)
    dg = (logits[:1].cpu() - ref_logits).abs().max().item()
    dl = (lat[:1].cpu() - ref_lat).abs().max().item()
    print(f"fp8 {w}x{h}: max|dlatent| {dl:.3e}  max|dlogit| {dg:.3e}")
    assert dg <= 1e-2 and dl <= FP8_LATENT_MAX


def test_small_config_with_fused_shortcut_on_every_halo_tile_mode():
    """block_out_channels (64, 128): a 128-cout conv2 with the fused 1x1 shortcut (64 -> 128).  Every value of flag 3
    must pick the same tile for the launch and for the GroupNorm-partials bookkeeping (a mismatch reads stale partials)."""
    from vae_tagger_amd.autoencoder_kl import AutoencoderKL
    cfg = dict(block_out_channels=(64, 128), down_block_types=("DownEncoderBlock2D",) * 2, latent_channels=16,
               use_quant_conv=False, scaling_factor=0.3611, shift_factor=0.1159)
    m = AutoencoderKL(**cfg)
    sd = synth.synth_state_dict(synth.encoder_manifest((64, 128)), seed=2)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    m = m.to("cuda").eval()
    x = synth.synth_images(2, 96, 80, seed=4)
    ref = encoder_ref.encoder_moments(sd, x, n_down=2)
    ctx = m._context()
    try:
        for occ2 in (0, 1, 2, 3, 4):                          # (4: the one-wave-per-SIMD experiment tile)
            for fuse_sc in (1, 0):
                ctx.call("vt_set_flag", 3, occ2)
                ctx.call("vt_set_flag", 8, fuse_sc)
                got = m.encode(x.cuda()).latent_dist.parameters.cpu()
                err = (got - ref).abs().max().item()
                assert err <= 3e-2, (occ2, fuse_sc, err)       # un-scaled moments: 1e-2 / 0.3611
    finally:
        ctx.call("vt_set_flag", 3, 3)
        ctx.call("vt_set_flag", 8, 1)


def test_device_preprocess_is_bit_exact_with_totensor_normalize(vae):
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    pipe = EncodeTagPipeline(vae, _decoder(11))
    u8 = torch.randint(0, 256, (2, 37, 53, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    u8[0, 0, 0] = torch.tensor([0, 127, 255], dtype=torch.uint8)
    ref = (u8.permute(0, 3, 1, 2).to(torch.float32).div(255.0) - 0.5) / 0.5       # ToTensor + Normalize(0.5, 0.5)
    got = pipe.normalize_u8(u8.cuda()).cpu()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("gain,S", [(1.0, 200), (6.0, 200), (1.0, 1024), (6.0, 1024), (6.0, 333), (1.0, 64), (6.0, 64), (1.0, 2048), (6.0, 2048)])   # every GEMM tile config, plain and flagged; 1024 / 2048: Q.K^T key sweep split over 2 / 4 workgroups
def test_mid_attention_without_softmax_pass(gain, S):
    """vt_op_attention (E5): the default path has no softmax pass -- Q.K^T emits exp(s - c_i) with c_i from operand norms,
    P.V divides by the row sums.  Every mode stays on the fp32 reference; with to_q / to_k scaled by 6 the norm bound is
    too loose (u - l > 120), the launch group is flagged and c_i is the exact row maximum: bit-identical to mode 1."""
    import ctypes
    from vae_tagger_amd.diffusers_vae_loader import get_diffusers_vae_config, load_diffusers_vae_from_config
    from _util import vp
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    A = "encoder.mid_block.attentions.0."
    for k in ("to_q", "to_k"):
        sd[A + k + ".weight"] = sd[A + k + ".weight"] * gain
        sd[A + k + ".bias"] = sd[A + k + ".bias"] * gain
    m = load_diffusers_vae_from_config(get_diffusers_vae_config())
    m.load_state_dict(sd, strict=False)
    ctx = m.to("cuda").eval()._context()
    B, C = 3, 512
    g = torch.Generator().manual_seed(int(gain) * 1000 + S)
    x = torch.randn(B, S, C, generator=g).bfloat16()
    res = torch.randn(B, S, C, generator=g)
    bf = lambda w: w.bfloat16().float()
    q = bf(x.float() @ bf(sd[A + "to_q.weight"]).t() + sd[A + "to_q.bias"])
    k = bf(x.float() @ bf(sd[A + "to_k.weight"]).t() + sd[A + "to_k.bias"])
    v = bf(x.float() @ bf(sd[A + "to_v.weight"]).t() + sd[A + "to_v.bias"])
    o = torch.softmax(q @ k.transpose(1, 2) / C ** 0.5, dim=-1) @ v
    ref = o @ bf(sd[A + "to_out.0.weight"]).t() + sd[A + "to_out.0.bias"] + res
    ws = torch.empty(ctx.lib.vt_op_attention_workspace_bytes(B, S, C), dtype=torch.uint8, device="cuda")
    xd, rd = x.cuda(), res.cuda()
    try:
        for qk, pv in ((1, 1), (1, 0), (0, 0)):                     # dedicated Q.K^T (+ fragment-order P.V) kernels / generic GEMMs
            ctx.call("vt_set_flag", 9, qk)
            ctx.call("vt_set_flag", 12, pv)
            outs = []
            for mode in (0, 1, 2):
                ctx.call("vt_set_flag", 7, mode)
                out = torch.full((B, S, C), float("nan"), device="cuda")
                ctx.call("vt_op_attention", vp(xd), vp(rd), vp(out), B, S, C, vp(ws), ctypes.c_void_p(0))
                torch.cuda.synchronize()
                outs.append(out.cpu())
            for out in outs:
                assert (out - ref).abs().max().item() <= 2e-2       # bf16 P and o: ~4e-3 relative on |o| <= max|v|
            assert (outs[0] - outs[1]).abs().max().item() <= 1e-2
            assert torch.equal(outs[0], outs[1]) == (gain > 1)      # flagged <=> the exact-maximum path ran
            if qk and pv:
                # the q | k and v^T projections on attn_qk.hip's skeleton (flag 17, default) against the generic GEMM: the same bf16 products
                # summed in another order -- both on the reference, and within bf16 rounding of each other
                ctx.call("vt_set_flag", 7, 0)
                ctx.call("vt_set_flag", 17, 0)
                out = torch.full((B, S, C), float("nan"), device="cuda")
                ctx.call("vt_op_attention", vp(xd), vp(rd), vp(out), B, S, C, vp(ws), ctypes.c_void_p(0))
                torch.cuda.synchronize()
                ctx.call("vt_set_flag", 17, 1)
                assert (out.cpu() - ref).abs().max().item() <= 2e-2 and (out.cpu() - outs[0]).abs().max().item() <= 1e-2
    finally:
        ctx.call("vt_set_flag", 7, 0)
        ctx.call("vt_set_flag", 9, 1)
        ctx.call("vt_set_flag", 12, 1)
        ctx.call("vt_set_flag", 17, 1)


@pytest.mark.parametrize("gain,S", [(1.0, 200), (6.0, 200), (1.0, 1024), (3.0, 1024), (1.0, 333), (1.0, 64), (1.0, 2048), (3.0, 2500), (1.0, 130)])
def test_mid_block_attention_on_fp8_operands(gain, S):
    """fp8 mode (vt_set_flag 11; attn_fp8.hip): q | k, v^T and the softmax numerators as e4m3, both S x S contractions on
    v_mfma_scale_f32_16x16x128_f8f6f4.  Checked against fp32 attention of the SAME e4m3-rounded q, k, v (what the kernels multiply):
    the remaining difference is the e4m3 rounding of P (3 significand bits per numerator, averaged over the keys of a row) and the
    bf16 output.  Shapes: one / several 128-key tiles, ragged tails, key sweeps split over 2 / 4 workgroups (S = 1024, 2048), and
    gains that make rows peaky.  Mode 0 takes the exponent shift from a first sweep over every 4th / 8th key tile (all of them below
    16 tiles) and redoes a launch group with the exact maximum when a numerator passed e4m3's 448; vt_set_flag(7, 1) always takes the
    exact maximum.  The flag-14 switch puts the launch back on the bf16 kernels."""
    import ctypes
    from vae_tagger_amd.diffusers_vae_loader import get_diffusers_vae_config, load_diffusers_vae_from_config
    from _util import vp
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    A = "encoder.mid_block.attentions.0."
    for k in ("to_q", "to_k"):
        sd[A + k + ".weight"] = sd[A + k + ".weight"] * gain
        sd[A + k + ".bias"] = sd[A + k + ".bias"] * gain
    m = load_diffusers_vae_from_config(get_diffusers_vae_config())
    m.load_state_dict(sd, strict=False)
    ctx = m.to("cuda").eval()._context()
    B, C = 2, 512
    g = torch.Generator().manual_seed(int(gain) * 1000 + S)
    x = torch.randn(B, S, C, generator=g).bfloat16()
    res = torch.randn(B, S, C, generator=g)
    bf = lambda w: w.bfloat16().float()
    e4 = lambda t: (t * 8.0).clamp(-448, 448).to(torch.float8_e4m3fn).float() / 8.0
    ws = torch.empty(ctx.lib.vt_op_attention_workspace_bytes(B, S, C), dtype=torch.uint8, device="cuda")
    xd, rd = x.cuda(), res.cuda()
    for f15 in (1, 0):
        if f15:
            # flag 15 (default): the projections multiply e4m3 operands too -- tokens e4m3(8 x), [Wq; Wk] and Wv as e4m3(W / s) with one scale per
            # matrix -- and write q8 | k8 and v8^T directly (proj_fp8_kernel): no bf16 q | k / v^T in between
            def w8(*names):
                w = torch.cat([bf(sd[A + n + ".weight"]) for n in names])
                sc = w.abs().max() / 448.0
                return (w / sc).to(torch.float8_e4m3fn).float() * sc
            x8 = e4(x.float())
            wqk = w8("to_q", "to_k")
            q = e4(x8 @ wqk[:C].t() + sd[A + "to_q.bias"])
            k = e4(x8 @ wqk[C:].t() + sd[A + "to_k.bias"])
            v = e4(x8 @ w8("to_v").t() + sd[A + "to_v.bias"])
        else:
            q = e4(bf(x.float() @ bf(sd[A + "to_q.weight"]).t() + sd[A + "to_q.bias"]))
            k = e4(bf(x.float() @ bf(sd[A + "to_k.weight"]).t() + sd[A + "to_k.bias"]))
            v = e4(bf(x.float() @ bf(sd[A + "to_v.weight"]).t() + sd[A + "to_v.bias"]))
        o = torch.softmax(q @ k.transpose(1, 2) / C ** 0.5, dim=-1) @ v
        ref = o @ bf(sd[A + "to_out.0.weight"]).t() + sd[A + "to_out.0.bias"] + res
        outs = {}
        try:
            ctx.call("vt_set_flag", 11, 1)
            ctx.call("vt_set_flag", 15, f15)
            for name, f14, mode in (("fp8 mode 0", 1, 0), ("fp8 mode 1", 1, 1), ("bf16 kernels", 0, 0)):
                ctx.call("vt_set_flag", 14, f14)
                ctx.call("vt_set_flag", 7, mode)
                out = torch.full((B, S, C), float("nan"), device="cuda")
                ctx.call("vt_op_attention", vp(xd), vp(rd), vp(out), B, S, C, vp(ws), ctypes.c_void_p(0))
                torch.cuda.synchronize()
                outs[name] = out.cpu()
            st = ctx.status()
        finally:
            ctx.call("vt_set_flag", 7, 0)
            ctx.call("vt_set_flag", 14, 1)
            ctx.call("vt_set_flag", 15, 1)
            ctx.call("vt_set_flag", 11, 0)
        assert st == 0
        e0 = (outs["fp8 mode 0"] - ref).abs().max().item()
        e1 = (outs["fp8 mode 1"] - ref).abs().max().item()
        print(f"fp8 attention S={S} gain={gain} e4m3 projections={f15}: max|d| vs fp32 attention of the e4m3 operands: mode 0 {e0:.3e}, mode 1 {e1:.3e}; "
              f"rms {(outs['fp8 mode 0'] - ref).pow(2).mean().sqrt():.3e}")
        # flat rows (gain 1: the synthetic weights, u - l ~ 8) average the 3-bit rounding of P over hundreds of keys; rows that a few keys
        # dominate (gain 3, 6) keep more of e4m3's half step (2^-4 relative): observed 1.4 - 1.6e-2 against 1 - 5e-3 -- the price of e4m3 P,
        # bounded here, not hidden (with a shift bounded from operand norms instead of the exact maximum it was 3 - 7e-2)
        tol, tol_rms = (FP8_ATTN_TOL, 2e-3) if gain == 1.0 else (3e-2, 3e-3)
        rms = (outs["fp8 mode 0"] - ref).pow(2).mean().sqrt().item()
        assert torch.isfinite(outs["fp8 mode 0"]).all() and e0 <= tol and e1 <= tol and rms <= tol_rms
        same = torch.equal(outs["fp8 mode 0"], outs["fp8 mode 1"])
        print(f"   mode 0 (sampled maximum, exact redo if a numerator was clamped) == mode 1 (always exact): {same}")
        if (S + 127) // 128 < 16:
            assert same                    # fewer than 16 key tiles: the first sweep takes them all
        if S == 2500 and gain == 3.0 and not f15:
            assert same                    # peaky rows, 20 key tiles, every 4th sampled: a numerator passes 448, the flag is raised and the
                                           # two gated launches redo the group with the exact maximum (= what mode 1 always does)
        if S == 2048 and gain == 1.0:
            assert not same                # flat rows: the sampled shift holds, no redo (the bits differ from the exact-shift path's)
        assert torch.isfinite(outs["bf16 kernels"]).all() and not torch.equal(outs["bf16 kernels"], outs["fp8 mode 0"])
        if gain == 1.0:         # (flag 14 off: bf16 q, k, v, P -- the e4m3 rounding of the reference's operands is what shows here)
            assert (outs["bf16 kernels"] - ref).abs().max().item() <= 6e-2
        # flag 14 off in fp8 mode = the bf16 projections and kernels, for EVERY gain: against fp32 attention of the bf16 operands they multiply
        qb = bf(x.float() @ bf(sd[A + "to_q.weight"]).t() + sd[A + "to_q.bias"])
        kb = bf(x.float() @ bf(sd[A + "to_k.weight"]).t() + sd[A + "to_k.bias"])
        vb = bf(x.float() @ bf(sd[A + "to_v.weight"]).t() + sd[A + "to_v.bias"])
        ref_b = (torch.softmax(qb @ kb.transpose(1, 2) / C ** 0.5, dim=-1) @ vb) @ bf(sd[A + "to_out.0.weight"]).t() + sd[A + "to_out.0.bias"] + res
        eb = (outs["bf16 kernels"] - ref_b).abs().max().item()
        print(f"   flag 14 off (bf16 kernels inside fp8 mode) vs fp32 attention of the bf16 operands: {eb:.3e}")
        assert eb <= 2e-2


def test_evaluation_caller_matches_oracle(vae, tmp_path):
    """evaluate_model / find_optimal_threshold (reference evaluation.py:173-275) over a synthetic loader: probabilities come
    from the HIP path, the expected metrics from the CPU oracle's probabilities through the same evaluator."""
    import json
    from vae_tagger_amd.evaluation import MultiLabelEvaluator, evaluate_model, find_optimal_threshold
    n_tags = 11
    dec = _decoder(n_tags)
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n_tags), seed=1)
    g = torch.Generator().manual_seed(5)
    batches = []
    for i in range(3):
        x = synth.synth_images(2, 64, 64, seed=40 + i)
        batches.append({"pixel_values": x, "labels": (torch.rand(2, n_tags, generator=g) < 0.4).float()})
    names = [f"tag_{i:05d}" for i in range(n_tags)]
    probs = [torch.sigmoid(decoder_ref.attention_decoder_forward(sd_d, encoder_ref.vae_wrapper_encode(sd_e, b["pixel_values"])))
             for b in batches]
    # a threshold no oracle probability comes close to (the HIP logits are within 1e-2 of the oracle's, not equal)
    flat = torch.cat([p.flatten() for p in probs]).sort().values
    gaps = flat[1:] - flat[:-1]
    k = int(gaps.argmax())
    thr = float((flat[k] + flat[k + 1]) / 2)
    assert gaps[k] > 1e-2
    m = evaluate_model(vae, dec, batches, names, device="cuda", threshold=thr, output_dir=str(tmp_path))
    ref = MultiLabelEvaluator(names, "cpu")
    for b, p in zip(batches, probs):
        ref.update((p > thr).float(), b["labels"], p)
    want = ref.compute_metrics(thr)
    for k in ("accuracy", "hamming_loss", "f1_micro", "f1_macro", "precision_weighted", "recall_micro"):
        assert abs(m[k] - want[k]) < 1e-6, k
    assert abs(m["mAP"] - want["mAP"]) < 5e-2          # ranks of near-tied probabilities may swap within the bf16 tolerance
    assert json.load(open(tmp_path / "evaluation_results_overall.json"))["f1_micro"] == m["f1_micro"]
    assert (tmp_path / "evaluation_results.csv").read_text().splitlines()[0] == "class_name,precision,recall,f1,ap,support"
    r = find_optimal_threshold(vae, dec, batches, names, device="cuda", output_dir=str(tmp_path))
    assert 0.1 <= r["global_threshold"] < 0.9 and set(r["per_class_thresholds"]) == set(names)


def test_device_resize_is_bit_exact_with_pillow(vae):
    """vt_resize_u8 / EncodeTagPipeline.load_image against Pillow itself (the library the reference's transforms call,
    modules.py:126-178): uint8 results identical, and the normalised fp32 tensor identical to ToTensor + Normalize."""
    import numpy as np
    from PIL import Image
    from oracle import resize_ref
    from vae_tagger_amd.modules import get_image_transform
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    pipe = EncodeTagPipeline(vae, _decoder(11))
    rng = np.random.default_rng(3)
    for (h, w, ow, oh) in [(37, 53, 16, 16), (480, 640, 256, 256), (1333, 2000, 1024, 1024), (100, 90, 256, 320), (64, 64, 64, 200),
                           (200, 64, 64, 64), (50, 50, 50, 50)]:
        a = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        for filt, pk in ((pipe.FILTER_BILINEAR, Image.BILINEAR), (pipe.FILTER_LANCZOS, Image.LANCZOS)):
            want = np.asarray(Image.fromarray(a).resize((ow, oh), pk))
            got = pipe.resize_u8(a, ow, oh, filt).cpu().numpy()
            assert np.array_equal(got, want), (h, w, ow, oh, filt, int(np.abs(got.astype(int) - want.astype(int)).max()))
    # crop box + the two transforms of get_image_transform, through load_image
    a = rng.integers(0, 256, (300, 500, 3)).astype(np.uint8)
    img = Image.fromarray(a)
    box = resize_ref.smart_crop_box(500, 300, 192, 256)
    want = np.asarray(img.crop((box[0], box[1], box[0] + box[2], box[1] + box[3])).resize((192, 256), Image.LANCZOS))
    assert np.array_equal(pipe.resize_u8(a, 192, 256, pipe.FILTER_LANCZOS, box).cpu().numpy(), want)
    for kw, ref_t in (({"resolution": 128}, get_image_transform(128)), ({"bucket": (192, 256)}, get_image_transform(0, True, (192, 256)))):
        assert torch.equal(pipe.load_image(img, **kw).cpu(), ref_t(img))


@pytest.mark.parametrize("extra", [[], ["--fp8"], ["--bucketed", "--bucket-batch", "2"]], ids=["configs2", "configs4_fp8", "configs3_bucketed"])
def test_bench_two_ranks_on_one_gpu_rehearsal(tmp_path, extra):
    """The N > 1 control flow of bench.py with the HIP path under it (default, fp8 and bucketed workloads): two ranks, both on GPU 0, gloo for the collectives
    (VT_BENCH_REHEARSAL=1; RCCL needs one GPU per rank, which this box does not have).  Rank-distinct inputs, one all-gather of
    logits per step, max-over-ranks timing, ONE JSON line from rank 0."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VT_BENCH_REHEARSAL="1", PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(29600 + os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--height", "256", "--width", "256", "--tags", "1000"] + extra
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2"
    assert res["config"]["global_batch"] == (8 if "--bucketed" in extra else 4)        # bucketed: two same-shape batches of 2 per rank and step
    assert res["dtype"] == ("fp8" if "--fp8" in extra else "bf16")
    assert res["value"] > 0 and res["scaling"] == "weak" and "cpu_baseline" not in res


def test_rccl_one_rank_communicator_runs_the_logits_all_gather(tmp_path):
    """RCCL executes on this ROCm build: a child process creates a ONE-rank "nccl" (= RCCL) process group on cuda:0 before any
    other GPU call, runs the exact collective `bench.py --gpus N` uses -- sharding.all_gather_logits' all_gather_into_tensor
    branch -- on a [16, 10000] fp32 tensor, and a barrier.  It proves that the library loads, a communicator is created and the
    collective runs; it says NOTHING about scaling over xGMI (no multi-GPU box is available to this build)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "import torch, torch.distributed as dist\n"
        "dev = torch.device('cuda', 0)\n"
        "dist.init_process_group('nccl', world_size=1, rank=0, device_id=dev)\n"
        "from vae_tagger_amd import sharding\n"
        "assert dist.get_backend() == 'nccl'\n"
        "x = torch.arange(16 * 10000, dtype=torch.float32, device=dev).reshape(16, 10000)\n"
        "y = sharding.all_gather_logits(x, [16], force_collective=True)\n"
        "dist.barrier()\n"
        "torch.cuda.synchronize()\n"
        "assert y.data_ptr() != x.data_ptr() and torch.equal(y, x)\n"
        "print('RCCL_OK', torch.cuda.nccl.version())\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29950 + os.getpid() % 40), PYTHONDONTWRITEBYTECODE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"RCCL_OK" in r.stdout, (r.stdout.decode()[-500:], r.stderr.decode()[-2000:])
    print(r.stdout.decode().strip())


def test_config4_fp8_batch16_1024_matches_oracle_and_is_batch_invariant(vae):
    """BASELINE configs[4] at its per-GPU batch: fp8 mode, 16 x 1024^2, 10 000 tags (16-image fp8 halo grids, the mid-block
    attention as two launch groups).  Images 0 and 11 against the CPU oracle (north_star's fp8 line: logits within 1e-2), and the
    mode is batch-invariant bit for bit: logits(x[4:13]) == logits(x)[4:13]."""
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    n = 10000
    pipe = EncodeTagPipeline(vae, _decoder(n))
    x = synth.synth_images(16, 1024, 1024, seed=1000)
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n), seed=1)
    xg = x.cuda()
    try:
        pipe.ctx.call("vt_set_flag", 11, 1)
        logits, lat = pipe.logits(xg, return_latent=True)
        part = pipe.logits(xg[4:13])
    finally:
        pipe.ctx.call("vt_set_flag", 11, 0)
    assert pipe.status() == 0
    assert torch.equal(part, logits[4:13])
    for i in (0, 11):
        ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x[i:i + 1])
        ref_logits = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
        dl = lat[i:i + 1].cpu() - ref_lat
        dg = (logits[i:i + 1].cpu() - ref_logits).abs().max().item()
        print(f"fp8 batch 16, image {i}: max|dlatent| {dl.abs().max():.3e} rms {dl.pow(2).mean().sqrt():.3e}  max|dlogit| {dg:.3e}")
        assert dg <= 1e-2
        _check_tag_order(pipe, logits[i:i + 1], ref_logits, f"configs[4] fp8 image {i}")
        assert dl.abs().max().item() <= FP8_LATENT_MAX and dl.pow(2).mean().sqrt().item() <= FP8_LATENT_RMS


@pytest.mark.parametrize("res,gain", [(256, 3.0), (256, 6.0), (512, 3.0), (512, 6.0)])
def test_config4_fp8_with_peaky_attention_rows_through_to_the_logits(res, gain):
    """fp8 mode on a checkpoint whose mid-block attention rows are dominated by a few keys (to_q / to_k x 3 and x 6: the weights
    test_mid_block_attention_on_fp8_operands bounds at the operator), through the rest of the encoder and the decoder to the
    LOGITS, against the oracle run on the same scaled state-dict (the diffusers Attention behind diffusers_vae_loader.py:79).
    north_star's fp8 line: logits within 1e-2; the status word stays clear; the tag order agrees wherever it is decidable."""
    from vae_tagger_amd.diffusers_vae_loader import (DiffusersVAEWrapper, get_diffusers_vae_config,
                                                      load_diffusers_vae_from_config)
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    n = 10000
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    A = "encoder.mid_block.attentions.0."
    for k in ("to_q", "to_k"):
        sd_e[A + k + ".weight"] = sd_e[A + k + ".weight"] * gain
        sd_e[A + k + ".bias"] = sd_e[A + k + ".bias"] * gain
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n), seed=1)
    m = load_diffusers_vae_from_config(get_diffusers_vae_config())
    m.load_state_dict(sd_e, strict=False)
    w = DiffusersVAEWrapper(m).to("cuda").eval()
    pipe = EncodeTagPipeline(w, _decoder(n))
    x = synth.synth_images(2 if res < 512 else 1, res, res, seed=int(gain) * 10 + res)
    ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x)
    ref_logits = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
    bf16_logits = pipe.logits(x.cuda())
    assert pipe.status() == 0
    try:
        pipe.set_fp8(True)
        logits, lat = pipe.logits(x.cuda(), return_latent=True)
        st = pipe.status()
        pipe.ctx.call("vt_set_flag", 7, 1)                        # always the exact row maximum
        logits_exact = pipe.logits(x.cuda())
        st_exact = pipe.status()
    finally:
        pipe.ctx.call("vt_set_flag", 7, 0)
        pipe.set_fp8(False)
    dg = (logits.cpu() - ref_logits).abs().max().item()
    dge = (logits_exact.cpu() - ref_logits).abs().max().item()
    db = (bf16_logits.cpu() - ref_logits).abs().max().item()
    dl = lat.cpu() - ref_lat
    print(f"fp8, to_q/to_k x {gain}, {res}^2: max|dlogit| {dg:.3e} (exact row maximum: {dge:.3e}; bf16 path: {db:.3e}); "
          f"max|dlatent| {dl.abs().max():.3e} rms {dl.pow(2).mean().sqrt():.3e}; vt_status {st} / {st_exact}")
    assert st == 0 and st_exact == 0
    assert db <= 1e-3
    assert dg <= 1e-2 and dge <= 1e-2
    for i in range(x.shape[0]):
        _check_tag_order(pipe, logits[i:i + 1], ref_logits[i:i + 1], f"fp8 peaky x{gain} {res}^2 image {i}")


@pytest.mark.parametrize("f15", [1, 0])
def test_fp8_attention_clamped_v_raises_the_saturation_bit(f15):
    """|8 v| > 448 cannot be represented in e4m3: with e4m3 projections (flag 15) proj_fp8_kernel raises VT_STATUS_FP8_SATURATED, with bf16
    projections the v^T conversion pass (attn_vt_to_fp8_kernel) does -- a clamped attention is never returned silently."""
    import ctypes
    from vae_tagger_amd._lib import VT_STATUS_FP8_SATURATED
    from vae_tagger_amd.diffusers_vae_loader import get_diffusers_vae_config, load_diffusers_vae_from_config
    from _util import vp
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    A = "encoder.mid_block.attentions.0."
    sd[A + "to_v.bias"] = sd[A + "to_v.bias"] + 80.0
    m = load_diffusers_vae_from_config(get_diffusers_vae_config())
    m.load_state_dict(sd, strict=False)
    ctx = m.to("cuda").eval()._context()
    B, S, C = 1, 200, 512
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, S, C, generator=g).bfloat16().cuda()
    res = torch.randn(B, S, C, generator=g).cuda()
    ws = torch.empty(ctx.lib.vt_op_attention_workspace_bytes(B, S, C), dtype=torch.uint8, device="cuda")
    out = torch.empty(B, S, C, device="cuda")
    try:
        ctx.call("vt_set_flag", 11, 1)
        ctx.call("vt_set_flag", 15, f15)
        ctx.call("vt_op_attention", vp(x), vp(res), vp(out), B, S, C, vp(ws), ctypes.c_void_p(0))
        assert ctx.status() == VT_STATUS_FP8_SATURATED and ctx.status() == 0
        ctx.call("vt_set_flag", 11, 0)
        ctx.call("vt_op_attention", vp(x), vp(res), vp(out), B, S, C, vp(ws), ctypes.c_void_p(0))
        assert ctx.status() == 0 and torch.isfinite(out).all()
    finally:
        ctx.call("vt_set_flag", 15, 1)
        ctx.call("vt_set_flag", 11, 0)


def _smooth_images(b, h, w, seed):
    """Pictures, not noise: a 12 x 12 random field upsampled bicubically + a little noise, clamped and quantised to 8 bits like a decoded file,
    then ToTensor + Normalize(0.5, 0.5).  Flat and saturated regions make neighbouring pixels' rounding errors EQUAL, so a 3x3 conv adds them
    coherently: bf16 operands alone are 1e-2 .. 2e-2 from the fp32 oracle on such inputs (7e-3 on uniform noise)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(b, 3, 12, 12, generator=g)
    img = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False)
    img = ((img + 0.06 * torch.randn(b, 3, h, w, generator=g)).clamp(0, 1) * 255).to(torch.uint8)
    return (img.float() / 255.0 - 0.5) / 0.5


@pytest.mark.parametrize("h,w,smooth", [(256, 256, True), (512, 512, True), (256, 256, False), (100, 148, True), (576, 768, True)])
def test_fp16_operand_mode_meets_the_tolerance_on_pictures(vae, h, w, smooth):
    """vt_set_flag(ctx, 18, 1): fp16 instead of bf16 MFMA operands for every convolution (halo, stride-2 and conv_out kernels on
    v_mfma_f32_16x16x32_f16; GroupNorm outputs, operand copies and weights as fp16 bits).  On SMOOTH inputs the bf16 path is 1e-2 .. 2e-2 from
    the fp32 oracle in the latent maximum (printed; the oracle's own bf16-operand emulation says the same) -- fp16 operands are inside
    north_star's 1e-2 with room to spare: asserted <= 4e-3 (observed ~1.5e-3 .. 2.5e-3), logits <= 1e-3; deterministic; status word clear;
    ragged and bucket shapes included.  The attention stays bf16."""
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    n = 1000
    pipe = EncodeTagPipeline(vae, _decoder(n))
    x = _smooth_images(2, h, w, seed=h * 3 + w) if smooth else synth.synth_images(2, h, w, seed=5)
    sd_e = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    sd_d = synth.synth_state_dict(synth.attention_decoder_manifest(n), seed=1)
    ref_lat = encoder_ref.vae_wrapper_encode(sd_e, x[:1])
    ref_logits = decoder_ref.attention_decoder_forward(sd_d, ref_lat)
    lg_b, lat_b = pipe.logits(x.cuda(), return_latent=True)
    try:
        pipe.set_fp16_operands(True)
        lg_h, lat_h = pipe.logits(x.cuda(), return_latent=True)
        again = pipe.logits(x.cuda())
        part = pipe.logits(x[1:].cuda())
    finally:
        pipe.set_fp16_operands(False)
    assert pipe.status() == 0 and torch.equal(again, lg_h) and torch.equal(part, lg_h[1:])
    db = (lat_b[:1].cpu() - ref_lat).abs().max().item()
    dh = (lat_h[:1].cpu() - ref_lat)
    gb = (lg_b[:1].cpu() - ref_logits).abs().max().item()
    gh = (lg_h[:1].cpu() - ref_logits).abs().max().item()
    print(f"{'smooth' if smooth else 'noise'} {w}x{h}: max|dlatent| bf16 operands {db:.3e} -> fp16 operands {dh.abs().max():.3e} (rms {dh.pow(2).mean().sqrt():.3e}); "
          f"max|dlogit| {gb:.3e} -> {gh:.3e}")
    assert dh.abs().max().item() <= 4e-3 and gh <= 1e-3
    assert torch.equal(pipe.logits(x.cuda()), lg_b)                     # back on bf16 operands: the default path's bits
    _check_tag_order(pipe, lg_h[:1], ref_logits, f"fp16 operands {w}x{h}")


def test_fp16_operand_mode_on_a_config_without_halo_tiles():
    """block_out_channels (64, 128): the 64-cout convs have no halo tile (Cout % 128 != 0) and stay on the bf16 generic GEMM; the 128-cout ones,
    the fused 64 -> 128 shortcut, the stride-2 conv and conv_out take fp16 operands -- every 16-bit operand tensor must carry the type its own
    consumer expects (a mismatch reads fp16 bits as bf16: errors of order 1)."""
    from vae_tagger_amd.autoencoder_kl import AutoencoderKL
    cfg = dict(block_out_channels=(64, 128), down_block_types=("DownEncoderBlock2D",) * 2, latent_channels=16,
               use_quant_conv=False, scaling_factor=0.3611, shift_factor=0.1159)
    m = AutoencoderKL(**cfg)
    sd = synth.synth_state_dict(synth.encoder_manifest((64, 128)), seed=2)
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()
    x = synth.synth_images(2, 96, 80, seed=4)
    ref = encoder_ref.encoder_moments(sd, x, n_down=2)
    ctx = m._context()
    try:
        for fuse_sc in (1, 0):
            for s2 in (1, 0):
                ctx.call("vt_set_flag", 8, fuse_sc)
                ctx.call("vt_set_flag", 13, s2)
                m.set_fp16_operands(False)
                e_b = (m.encode(x.cuda()).latent_dist.parameters.cpu() - ref).abs().max().item()
                m.set_fp16_operands(True)
                e_h = (m.encode(x.cuda()).latent_dist.parameters.cpu() - ref).abs().max().item()
                print(f"(64, 128) config, fused shortcut {fuse_sc}, stride-2 halo {s2}: max|dmoments| bf16 {e_b:.3e}, fp16 operands {e_h:.3e}")
                assert e_b <= 3e-2 and e_h <= 3e-2 and m.status() == 0
    finally:
        ctx.call("vt_set_flag", 8, 1)
        ctx.call("vt_set_flag", 13, 1)
        m.set_fp16_operands(False)


@pytest.mark.parametrize("mode", ["bf16", "fp8", "fp16_operands"])
def test_planar_downsample_input_is_bit_identical(vae, mode):
    """vt_set_flag(ctx, 19, v): the copy of a stage's output that feeds its stride-2 conv, written chunk-planar ([C/32][H][W][32]; e4m3:
    [C/64][H][W][64]) by the producing conv2 and gathered from there by the phase-plane kernel (default), against NHWC (0): the same values in
    another place -- latents identical bit for bit in every numeric mode, on even, odd and ragged sizes."""
    ctx = vae.vae._context()
    old_check = vae.check_finite
    vae.check_finite = False
    try:
        if mode == "fp8":
            ctx.call("vt_set_flag", 11, 1)
        if mode == "fp16_operands":
            ctx.call("vt_set_flag", 18, 1)
        for (b, hh, ww) in ((2, 128, 192), (1, 100, 76), (2, 72, 88), (1, 264, 136)):
            x = synth.synth_images(b, hh, ww, seed=hh + ww).cuda()
            ctx.call("vt_set_flag", 19, 1)
            a = vae.encode(x)
            ctx.call("vt_set_flag", 19, 0)
            bb = vae.encode(x)
            assert torch.isfinite(a).all() and torch.equal(a, bb), (mode, hh, ww, (a - bb).abs().max().item())
        assert ctx.status() == 0
    finally:
        ctx.call("vt_set_flag", 19, 1)
        ctx.call("vt_set_flag", 11, 0)
        ctx.call("vt_set_flag", 18, 0)
        vae.check_finite = old_check


@pytest.mark.parametrize("h,w", [(64, 64), (72, 88), (128, 192), (264, 136), (512, 512)])
def test_conv_out_on_its_halo_tile_matches_the_generic_gemm(vae, h, w):
    """vt_set_flag(ctx, 20, v): conv_out (512 -> 32, the moments / mode() epilogue) on conv_out_halo.hip's 32-cout halo tile
    (default) against the generic GEMM (0): the same bf16 products summed in another order -- moments (all 32 channels) and the scaled mode()
    (first 16, * 0.3611 + 0.1159) within 2e-3 of each other, both within tolerance of the oracle; latent sizes that are not multiples of
    the 16 x 16 tile (9 x 11, 33 x 17) and one below a tile (8 x 8); fp16-operand form included."""
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    x = synth.synth_images(2, h, w, seed=h + 2 * w)
    ref_m = encoder_ref.encoder_moments(sd, x) if h * w <= 128 * 192 else None
    ctx = vae.vae._context()
    xd = x.cuda()
    try:
        for f16 in (0, 1):
            ctx.call("vt_set_flag", 18, f16)
            ctx.call("vt_set_flag", 20, 0)
            m0 = vae.vae.encode(xd).latent_dist.parameters.clone()
            z0 = vae.encode(xd).clone()
            ctx.call("vt_set_flag", 20, 1)
            m1 = vae.vae.encode(xd).latent_dist.parameters
            z1 = vae.encode(xd)
            assert m1.shape == m0.shape == (2, 32, h // 8, w // 8) and z1.shape == (2, 16, h // 8, w // 8)
            dm, dz = (m1 - m0).abs().max().item(), (z1 - z0).abs().max().item()
            print(f"{w}x{h} fp16 operands {f16}: conv_out halo tile vs generic GEMM: max|dmoments| {dm:.2e}, max|dlatent| {dz:.2e}")
            assert torch.isfinite(m1).all() and dm <= 2e-3 and dz <= 1e-3
            assert torch.allclose(m1[:, :16] * 0.3611 + 0.1159, z1, atol=1e-6)
            assert torch.equal(vae.encode(xd), z1)
            if ref_m is not None:
                assert (m1.cpu() - ref_m).abs().max().item() <= 3e-2     # un-scaled moments: 1e-2 / 0.3611
    finally:
        ctx.call("vt_set_flag", 18, 0)
        ctx.call("vt_set_flag", 20, 1)
    assert ctx.status() == 0


@pytest.mark.parametrize("h,w", [(72, 88), (128, 192), (264, 136), (512, 512)])
def test_attention_linear_layers_on_the_qk_skeleton_match_the_generic_gemm(vae, h, w):
    """vt_set_flag(ctx, 17, v): the attention's q | k, v^T and to_out linear layers on attn_qk.hip's skeleton (modes 4 / 5: one operand's rows in
    registers, the other's streamed; to_out with the residual add, fp16 / fp32 stores and the GroupNorm partials of the next norm in its
    epilogue) against the generic GEMM (0), through the whole encoder: token counts that are not multiples of 32 / 64 / 256 (99, 561), both
    residual storage types (flag 4), latents within 2e-3 of each other and within tolerance of the oracle, deterministic, status clear."""
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    x = synth.synth_images(2, h, w, seed=3 * h + w)
    ref = encoder_ref.vae_wrapper_encode(sd, x) if h * w <= 264 * 136 else None
    ctx = vae.vae._context()
    xd = x.cuda()
    try:
        for res16 in (1, 0):
            ctx.call("vt_set_flag", 4, res16)
            ctx.call("vt_set_flag", 17, 0)
            z0 = vae.encode(xd).clone()
            ctx.call("vt_set_flag", 17, 1)
            z1 = vae.encode(xd)
            d = (z1 - z0).abs().max().item()
            print(f"{w}x{h} (S = {(h // 8) * (w // 8)}), fp16 residual storage {res16}: linear layers on the Q.K^T skeleton vs generic GEMM: max|dlatent| {d:.2e}")
            assert torch.isfinite(z1).all() and d <= 2e-3 and torch.equal(vae.encode(xd), z1)
            if ref is not None:
                assert (z1.cpu() - ref).abs().max().item() <= TOL_LATENT_BF16
    finally:
        ctx.call("vt_set_flag", 4, 1)
        ctx.call("vt_set_flag", 17, 1)
    assert ctx.status() == 0


@pytest.mark.parametrize("mode", ["bf16", "fp8", "fp16_operands"])
def test_encoder_is_bit_stable_run_to_run_on_ragged_shapes(vae, mode):
    """Round 4: inputs whose latent sides are 1 mod 8 (136 x 264, 200 x 328, 264 x 520: ragged 16-pixel tiles on every level) encoded to
    latents that differed in up to every repeat at batch 4 (1.3e-2 apart in the moments): the GroupNorm partials of the halo conv's epilogue
    were not bit-stable (a packed-fp32 op_sel hazard).  The soak runs on 1024^2 never see a ragged tile.  Every numeric mode,
    25 repeats per shape, every latent bit equal to the first run's."""
    ctx = vae.vae._context()
    old_check = vae.check_finite
    vae.check_finite = False
    try:
        if mode == "fp8":
            ctx.call("vt_set_flag", 11, 1)
        if mode == "fp16_operands":
            ctx.call("vt_set_flag", 18, 1)
        for (b, hh, ww) in ((4, 264, 136), (2, 520, 264), (4, 136, 264), (2, 328, 200), (3, 100, 76)):
            x = synth.synth_images(b, hh, ww, seed=hh + 2 * ww).cuda()
            ref = vae.encode(x).clone()
            assert torch.isfinite(ref).all()
            bad = sum(int(not torch.equal(vae.encode(x), ref)) for _ in range(25))
            assert bad == 0, (mode, b, hh, ww, bad)
        assert ctx.status() == 0
    finally:
        ctx.call("vt_set_flag", 11, 0)
        ctx.call("vt_set_flag", 18, 0)
        vae.check_finite = old_check


@pytest.mark.parametrize("mode", ["bf16", "fp8", "fp16_operands"])
def test_c_abi_writes_stay_inside_the_buffers_it_was_given(vae, mode):
    """vt_encode_tag through the C ABI with guard bands: the workspace is exactly vt_encode_tag_workspace_bytes() long and the latent / logit buffers
    exactly their documented sizes (include/vae_tagger_hip.h), each followed by 1 MB of a byte pattern that must survive the call -- on tile-aligned,
    ragged and odd shapes, in every numeric mode.  (Round 4: a test that passed an 8 x 8 latent buffer for a 32 x 32 latent went unnoticed for rounds;
    this is the same check pointed at the library's own planner, which grew this round: to_out's partials, planar and fp16 operand copies.)"""
    import ctypes
    from vae_tagger_amd.pipeline import EncodeTagPipeline
    n = 37
    pipe = EncodeTagPipeline(vae, _decoder(n))
    pipe.check_finite = False
    ctx, L = pipe.ctx, pipe.ctx.lib
    GUARD, PAT = 1 << 20, 0xA5
    try:
        if mode == "fp8":
            pipe.set_fp8(True)
        if mode == "fp16_operands":
            ctx.call("vt_set_flag", 18, 1)
        for (b, hh, ww) in ((2, 128, 192), (3, 264, 136), (1, 100, 76), (2, 72, 88), (1, 512, 512)):
            x = synth.synth_images(b, hh, ww, seed=hh + ww).cuda()
            need = L.vt_encode_tag_workspace_bytes(ctx.handle, b, hh, ww)
            assert need > 0
            n_lat, n_log = b * 16 * (hh // 8) * (ww // 8) * 4, b * n * 4
            bufs = []
            for nbytes in (need, n_lat, n_log):
                t = torch.full((nbytes + 256 + GUARD,), PAT, dtype=torch.uint8, device="cuda")
                p = (t.data_ptr() + 255) // 256 * 256
                bufs.append((t, p, p - t.data_ptr()))
            (tw, pw, ow), (tl, pl, ol), (tg, pg, og) = bufs
            rc = L.vt_encode_tag(ctx.handle, ctypes.c_void_p(x.data_ptr()), b, hh, ww, ctypes.c_void_p(pl), ctypes.c_void_p(pg), ctypes.c_void_p(pw), need, None)
            assert rc == 0, L.vt_last_error(ctx.handle)
            torch.cuda.synchronize()
            for what, (t, p, off), nbytes in (("workspace", bufs[0], need), ("latent", bufs[1], n_lat), ("logits", bufs[2], n_log)):
                tail = t[off + nbytes:]
                assert bool((tail == PAT).all()), (mode, b, hh, ww, what, int((tail != PAT).nonzero()[0]))
                assert bool((t[:off] == PAT).all()), (mode, b, hh, ww, what, "bytes in front of the buffer")
            lat = tl[ol:ol + n_lat].view(torch.float32).view(b, 16, hh // 8, ww // 8)
            lg = tg[og:og + n_log].view(torch.float32).view(b, n)
            assert torch.isfinite(lat).all() and torch.isfinite(lg).all()
            ref_lg, ref_lat = pipe.logits(x, return_latent=True)                      # the Python mirror allocates by its own rule: same bits
            assert torch.equal(ref_lg, lg) and torch.equal(ref_lat, lat)
        assert pipe.status() == 0
    finally:
        pipe.set_fp8(False)
        ctx.call("vt_set_flag", 18, 0)
