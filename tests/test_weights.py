"""Pretrained-weight loading (feddat_amd/weights.py) against what the REFERENCE makes of the same files.

Fixtures: tests/golden/g13_vilt_pretrained.npz (HF `ViltModel.from_pretrained` + the reference's
ViltEncoderWrapper.expand_modality_type_embeddings + ViltContinualLearner; oracle/make_golden.py --only-g13) and
g14_albef_pretrained.npz (the reference's own `load_albef`; oracle/make_albef_golden.py --only-g14).  The checkpoint files
themselves (150-250 MB) are regenerated bit-identically by tests/ckpt_util.py from key-seeded fills.  CPU tests: every loaded
tensor vs the reference's state dict, and the CPU oracle's forward on the loaded parameters vs the reference's logits.
(GPU: tests/test_weights_gpu.py runs the HIP engine on the same parameters.)"""
import os

import numpy as np
import pytest
import torch

from oracle import albef_oracle as A
from oracle import feddat_oracle as O
from tests import ckpt_util
from tests.golden_util import load

ALBEF_DIMS = dict(vit_depth=1, enc_layers=7, fusion_layer=6, dec_layers=1, image=64, vocab=3072, max_pos=64)


# the reference's state dict lists the tied LM-head tensors under their own names too (xbert.py BertLMHeadModel)
TIED = {"text_decoder.cls.predictions.decoder.weight": "text_decoder.bert.embeddings.word_embeddings.weight",
        "text_decoder.cls.predictions.decoder.bias": "text_decoder.cls.predictions.bias"}


def _check_tensors(rec, sd):
    keys = [k[len("norm::"):] for k in rec if k.startswith("norm::")]
    assert len(keys) > 30
    for k in keys:
        src = k
        for a, b in TIED.items():
            if k.endswith(a):
                src = k[:-len(a)] + b
        assert src in sd, f"{k} not produced by the loader"
        f = sd[src].detach().float().flatten()
        assert abs(float(f.norm()) - float(rec["norm::" + k])) <= 1e-5 * float(rec["norm::" + k]) + 1e-7, k
        idx = (torch.arange(8, dtype=torch.int64) * (f.numel() - 1)) // 7
        np.testing.assert_allclose(f[idx].numpy(), rec["samp::" + k], rtol=1e-6, atol=1e-7, err_msg=k)
    return keys


@pytest.fixture(scope="module")
def vilt_dir(tmp_path_factory):
    return ckpt_util.write_hf_vilt_checkpoint(str(tmp_path_factory.mktemp("vilt_b32_mlm")), layers=2)


def vilt_params_from(path):
    from feddat_amd import weights
    P = weights.load_vilt_pretrained(path, ["art"], layers=2, seed=3)
    d = O.ViltDims(layers=2)
    for k, shp in O.param_shapes(d, ["art"]).items():          # the fixture's adapters / head are name-seeded
        if "adapter_" in k or k.startswith("task_layer."):
            assert tuple(P[k].shape) == tuple(shp), k
            P[k] = O.seeded_value(k, shp, 0.02, 0.02)
    return P, d


def test_vilt_hf_directory_loads_like_the_reference(vilt_dir, golden_dir):
    rec = load(golden_dir, "g13_vilt_pretrained.npz")
    from feddat_amd import weights
    raw = weights.load_vilt_pretrained(vilt_dir, ["art", "gqa"], layers=2, seed=3)
    keys = _check_tensors(rec, raw)
    # 2 -> 3 modality rows, row 2 a copy of row 1 (vilt.py:102-113); ViltOutput.dense renamed under Adaptered_ViltOutput
    tt = raw["vilt_encoder.vilt.embeddings.token_type_embeddings.weight"]
    assert tt.shape == (3, 768) and torch.equal(tt[2], tt[1]) and not torch.equal(tt[0], tt[1])
    assert any(".output.layer.dense.weight" in k for k in keys)
    # fresh trainables the reference way: adapters N(0, 0.02) / zero bias (adapter.py:5-14), heads nn.Linear defaults
    wd = raw["vilt_encoder.vilt.encoder.layer.1.output.adapter.adapter_1_down.weight"]
    assert 0.015 < float(wd.std()) < 0.025 and float(raw[
        "vilt_encoder.vilt.encoder.layer.1.output.adapter.adapter_1_down.bias"].abs().max()) == 0.0
    for t in ("art", "gqa"):
        w0 = raw[f"task_layer.{t}.clf_fc0.weight"]
        assert w0.shape == (1536, 768) and float(w0.abs().max()) <= 1 / 768 ** 0.5 + 1e-7
        assert torch.equal(raw[f"task_layer.{t}.clf_norm0.weight"], torch.ones(1536))
    assert not torch.equal(raw["task_layer.art.clf_fc0.weight"], raw["task_layer.gqa.clf_fc0.weight"])
    # the oracle's forward on the loaded parameters = the reference's forward on its loaded model
    P, d = vilt_params_from(vilt_dir)
    batch = O.synthetic_batch(2, 384, 1300)
    with torch.no_grad():
        for mode in ("gating", "adapter_1"):
            pooled, lg = O.vilt_forward(P, d, batch, mode, "art")
            assert float((pooled - torch.from_numpy(rec[f"fwd.{mode}.pooled"])).abs().max()) < 2e-5, mode
            assert float((lg - torch.from_numpy(rec[f"fwd.{mode}.logits"])).abs().max()) < 2e-5, mode


def test_vilt_bin_and_wrapper_state_dict_formats(tmp_path, vilt_dir):
    """pytorch_model.bin directories and a torch.save'd ViltEncoderWrapper state dict (`vilt.*` keys: the else-branch of
    load_vilt_encoder, vilt.py:407-417) give the same tensors as the safetensors directory."""
    from feddat_amd import weights
    from safetensors.torch import load_file
    ref = weights.convert_vilt_state_dict(weights.read_checkpoint(vilt_dir), 2)
    sd = dict(load_file(os.path.join(vilt_dir, "model.safetensors")))
    bin_dir = tmp_path / "bin"
    bin_dir.mkdir()
    torch.save(sd, str(bin_dir / "pytorch_model.bin"))
    wrapper = {k: v for k, v in sd.items() if k.startswith("vilt.")}          # ViltEncoderWrapper.state_dict()
    torch.save(wrapper, str(tmp_path / "encoder.pt"))
    bare = {k[len("vilt."):]: v for k, v in wrapper.items()}                  # ViltModel.state_dict()
    torch.save(bare, str(tmp_path / "viltmodel.pt"))
    for p in (str(bin_dir), str(tmp_path / "encoder.pt"), str(tmp_path / "viltmodel.pt")):
        got = weights.convert_vilt_state_dict(weights.read_checkpoint(p), 2)
        assert got.keys() == ref.keys()
        assert all(torch.equal(got[k], ref[k]) for k in ref), p
    with pytest.raises(weights.FeddatHipError):
        weights.load_vilt_pretrained(vilt_dir, ["art"], layers=3)             # the file has two layers


def test_missing_checkpoint_is_an_error_not_a_random_init(tmp_path):
    from feddat_amd import train, weights
    with pytest.raises(weights.FeddatHipError, match="no such local file"):
        weights.resolve("dandelin/vilt-b32-mlm")
    assert weights.resolve(None) is None
    with pytest.raises(weights.FeddatHipError):
        weights.read_checkpoint(str(tmp_path))                               # a directory without weights
    # main() refuses before it touches a device
    with pytest.raises(weights.FeddatHipError, match="no such local file"):
        train.main(["--encoder_name", "vilt", "--pretrained_model_name", "dandelin/vilt-b32-mlm", "--comm_rounds", "1"])


def test_interpolate_pos_embed_matches_torch_bicubic():
    """models/vit.py:193-217 uses F.interpolate(mode='bicubic', align_corners=False); the loader's host-side restatement."""
    import torch.nn.functional as F
    from feddat_amd import weights
    g = torch.Generator().manual_seed(0)
    for old, new in ((3, 4), (16, 24), (14, 7), (5, 5)):
        pos = torch.randn(1, old * old + 1, 96, generator=g)
        want = torch.cat([pos[:, :1], F.interpolate(pos[:, 1:].reshape(1, old, old, 96).permute(0, 3, 1, 2), size=(new, new),
                                                    mode="bicubic", align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)], 1)
        got = weights.interpolate_pos_embed(pos, new * new + 1)
        assert got.shape == want.shape and float((got - want).abs().max()) < 1e-5, (old, new)      # fp32 re-association only


@pytest.fixture(scope="module")
def albef_pth(tmp_path_factory):
    d = ALBEF_DIMS
    return ckpt_util.write_albef_checkpoint(str(tmp_path_factory.mktemp("albef") / "ALBEF.pth"), vit_depth=d["vit_depth"],
                                            enc_layers=d["enc_layers"], pre_image=48, vocab=d["vocab"], max_pos=d["max_pos"])


def albef_params_from(path):
    from feddat_amd import weights
    d = A.AlbefDims(**ALBEF_DIMS)
    P = weights.load_albef_pretrained(path, seed=1, **ALBEF_DIMS)
    for k, shp in A.param_shapes(d).items():
        assert k in P and tuple(P[k].shape) == tuple(shp), k
        if "adapter_" in k:
            P[k] = O.seeded_value(k, shp, 0.02, 0.02)
    return P, d


def test_albef_pth_loads_like_the_reference(albef_pth, golden_dir):
    rec = load(golden_dir, "g14_albef_pretrained.npz")
    P, d = albef_params_from(albef_pth)
    keys = _check_tensors(rec, P)
    # decoder layer 0 <- encoder layer 6 (cross-attention included), decoder embeddings / LM head <- the encoder's
    pre = A.PRE
    assert torch.equal(P[pre + "text_decoder.bert.encoder.layer.0.crossattention.self.key.weight"],
                       P[pre + "text_encoder.encoder.layer.6.crossattention.self.key.weight"])
    assert torch.equal(P[pre + "text_decoder.bert.embeddings.word_embeddings.weight"],
                       P[pre + "text_encoder.embeddings.word_embeddings.weight"])
    assert P[pre + "visual_encoder.pos_embed"].shape == (1, 17, 768)          # 3 x 3 grid of the file -> 4 x 4
    assert any("text_decoder.cls.predictions.transform" in k for k in keys)
    b0 = A.synthetic_batch(3, d, 1400, q_len=12, a_len=5, k=[2, 1, 3], ragged=True)
    with torch.no_grad():
        for mode in ("gating", "adapter_1"):
            loss, logits = A.albef_train_forward(P, d, b0, mode)
            assert abs(float(loss) - float(rec[f"fwd.{mode}.loss"])) < 2e-4 * float(rec[f"fwd.{mode}.loss"]), mode
            assert float((logits - torch.from_numpy(rec[f"fwd.{mode}.logits"])).abs().max()) < 2e-4, mode
