"""CPU: the oracle against the committed golden vectors (decoder goldens are outputs of the reference
itself, see oracle/make_goldens.py), plus the wiring checks SURVEY.md section 8c lists for the encoder."""
import pytest
import torch

from oracle import decoder_ref, encoder_ref
from vae_tagger_amd import synth

from _util import checksum, golden, latent_input

DEC_CASES = [
    ("attn_n11_16x16", 11, (2, 16, 16, 16), (True, True, False)),
    ("attn_n10000_64x64", 10000, (2, 16, 64, 64), (True, True, False)),
    ("attn_n11_72x128", 11, (1, 16, 72, 128), (True, True, False)),
    ("attn_cross_n11_16x16", 11, (2, 16, 16, 16), (True, True, True)),
    ("attn_nospatial_n11_16x16", 11, (2, 16, 16, 16), (False, True, False)),
]


@pytest.mark.parametrize("name,n,shape,flags", DEC_CASES)
def test_decoder_restatement_matches_reference_outputs(name, n, shape, flags):
    g = golden("decoder_" + name)
    sd = synth.synth_state_dict(synth.attention_decoder_manifest(n, 16, *flags), seed=1)
    x = latent_input(shape, seed=7)
    assert torch.allclose(checksum(x), g["input_checksum"], rtol=0, atol=1e-6), "input generator drifted"
    if "input" in g:
        assert torch.equal(x, g["input"])
    taps = {}
    logits = decoder_ref.attention_decoder_forward(sd, x, taps=taps)
    assert torch.allclose(logits, g["logits"], rtol=1e-5, atol=1e-5)
    assert torch.allclose(taps["compress"], g["after_compress"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(taps["self_attn"], g["after_self_attn"], rtol=1e-5, atol=1e-6)
    if "after_spatial" in g:
        assert torch.allclose(taps["spatial"], g["after_spatial"], rtol=1e-5, atol=1e-6)
    conf, idx = decoder_ref.get_confidence(logits)
    assert torch.allclose(conf, g["conf_sorted"], rtol=0, atol=1e-6)
    # argsort agrees with the reference's (unstable) sort wherever the neighbours are distinguishable
    gap = (g["conf_sorted"][:, :-1] - g["conf_sorted"][:, 1:]).abs()
    distinct = torch.ones_like(idx, dtype=torch.bool)
    distinct[:, :-1] &= gap > 1e-6
    distinct[:, 1:] &= gap > 1e-6
    assert torch.equal(idx[distinct], g["indices"][distinct])


def test_plain_decoder_restatement_matches_reference_outputs():
    g = golden("decoder_plain_n11_16x16")
    sd = synth.synth_state_dict(synth.plain_decoder_manifest(11), seed=1)
    logits = decoder_ref.plain_decoder_forward(sd, latent_input((2, 16, 16, 16), seed=7))
    assert torch.allclose(logits, g["logits"], rtol=1e-5, atol=1e-5)


def test_encoder_manifest_parameter_count_and_keys():
    m = synth.encoder_manifest()
    n = 0
    for s in m.values():
        k = 1
        for d in s:
            k *= d
        n += k
    assert n == 34_274_208           # FLUX AutoencoderKL encoder parameters
    assert m["encoder.conv_in.weight"] == (128, 3, 3, 3)
    assert m["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"] == (256, 128, 1, 1)
    assert m["encoder.down_blocks.2.resnets.0.conv_shortcut.weight"] == (512, 256, 1, 1)
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in m
    assert m["encoder.mid_block.attentions.0.to_out.0.weight"] == (512, 512)
    assert m["encoder.conv_out.weight"] == (32, 512, 3, 3)


@pytest.mark.parametrize("name,h,w", [("enc_64x64", 64, 64), ("enc_128x192", 128, 192)])
def test_encoder_restatement_matches_committed_latents(name, h, w):
    g = golden("encoder_" + name)
    sd = synth.synth_state_dict(synth.encoder_manifest(), seed=0)
    x = synth.synth_images(1, h, w, seed=3)
    assert torch.allclose(checksum(x), g["input_checksum"], rtol=0, atol=1e-6)
    lat = encoder_ref.vae_wrapper_encode(sd, x)
    assert lat.shape == (1, 16, h // 8, w // 8)      # agrees with get_vae_latent_info (modules.py:244-254)
    assert torch.allclose(lat, g["latent"], rtol=1e-4, atol=1e-5)


def test_encoder_flops_match_survey():
    assert abs(encoder_ref.encoder_flops(1024, 1024) / 1e12 - 4.8826) < 2e-3
    assert abs(encoder_ref.encoder_flops(512, 512) / 1e12 - 1.1176) < 2e-3


def test_resize_restatement_matches_pillow_bit_for_bit():
    """oracle/resize_ref.py restates Pillow's ImagingResample (what transforms.Resize / SmartResize end up calling,
    modules.py:126-178); Pillow itself is installed here and pins it: random images, up- and down-scaling, both filters."""
    import numpy as np
    from PIL import Image
    from oracle import resize_ref as R
    rng = np.random.default_rng(0)
    for _ in range(12):
        h, w = int(rng.integers(5, 160)), int(rng.integers(5, 160))
        oh, ow = int(rng.integers(4, 120)), int(rng.integers(4, 120))
        a = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        for kind, pk in ((R.BILINEAR, Image.BILINEAR), (R.LANCZOS, Image.LANCZOS)):
            want = np.asarray(Image.fromarray(a).resize((ow, oh), pk))
            assert np.array_equal(R.resize(a, ow, oh, kind), want), (h, w, oh, ow, kind)
    # SmartResize's crop box (centre mode) against the arithmetic at modules.py:150-175
    assert R.smart_crop_box(400, 200, 512, 512) == (100, 0, 200, 200)
    assert R.smart_crop_box(200, 400, 512, 512) == (0, 100, 200, 200)
    assert R.smart_crop_box(300, 300, 640, 640) == (0, 0, 300, 300)


def test_argsort_agreement_definition():
    """oracle/agreement.py (the metric's second half, reference modules.py:470-475 as consumed at infer_full.py:106-125): a
    perturbation of the logits inside +-d can only reorder tags whose oracle logits are within 2 d of each other, so the
    must-hold parts of the report hold by construction; a defect outside the band (two well-separated tags swapped in the
    index array, a tag dropped from the thresholded set) is caught."""
    from oracle.agreement import argsort_agreement
    g = torch.Generator().manual_seed(0)
    n = 10000
    ref = 0.4 * torch.randn(n, generator=g)
    for d in (2e-4, 5e-3):
        got = ref + d * (2 * torch.rand(n, generator=g) - 1)
        idx = decoder_ref.get_confidence(got[None])[1][0]
        a = argsort_agreement(ref, got, idx)
        assert a["identical_at_compared_ranks"] and a["swaps_stay_inside_the_2d_band"] and a["threshold_set_matches_outside_the_band"]
        assert 0 < a["ranks_compared"] < n and a["frac_identical_positions"] < 1.0 and a["top1_identical"] in (True, False)
        assert a["max_abs_dlogit"] <= d
    same = argsort_agreement(ref, ref, decoder_ref.get_confidence(ref[None])[1][0])
    assert same["frac_identical_positions"] == 1.0 and same["ranks_compared"] > 0.9 * n and same["max_rank_displacement"] == 0
    # a sort defect: the two top tags (far apart in the tail) exchanged
    got = ref + 2e-4 * (2 * torch.rand(n, generator=g) - 1)
    idx = decoder_ref.get_confidence(got[None])[1][0].clone()
    idx[[0, 40]] = idx[[40, 0]]
    a = argsort_agreement(ref, got, idx)
    assert not a["swaps_stay_inside_the_2d_band"] and not a["identical_at_compared_ranks"] and a["first_disagreeing_compared_rank"] == 0
    # a logit defect outside the tolerance on one tag shows in the thresholded set when d is stated by the caller
    got = ref.clone()
    k = int(torch.argmax((ref > 0.05).float()))
    got[k] = -1.0
    a = argsort_agreement(ref, got, decoder_ref.get_confidence(got[None])[1][0], max_abs_dlogit=1e-2)
    assert not a["threshold_set_matches_outside_the_band"]
    # ties: equal oracle logits are never "compared", and ascending index is the defined order
    tie = torch.tensor([0.5, 0.5, -1.0, 2.0])
    a = argsort_agreement(tie, tie, torch.tensor([3, 0, 1, 2]))
    assert a["frac_identical_positions"] == 1.0 and a["ranks_compared"] == 2
