"""Pretrained weights and the reference's RAW batch schema through the HIP engine.

  * G13 / G14: feddat_amd.weights reads the synthetic checkpoints of tests/ckpt_util.py (HF ViLT directory, ALBEF.pth); the
    engines built on the loaded parameters reproduce the logits the REFERENCE computes after loading the same files
    (ViltModel.from_pretrained + ViltEncoderWrapper / load_albef).
  * train.main(--pretrained_model_name <dir>) trains on the loaded backbone and refuses a path that does not exist.
  * model(task_key, images=[uint8 arrays], texts=[str]) and train_step(batch={"images", "raw_texts", "target_scores"})
    (vilt.py:87-100,244-264,455-459; task_trainer.py:248-264): device image processor + device WordPiece tokenizer once per
    batch, on the G7 image generator and the G9 vocabulary / questions, against the oracle's processor, tokenizer and model."""
import numpy as np
import pytest
import torch

from oracle import albef_oracle as A
from oracle import feddat_oracle as O
from oracle import image_oracle as IO
from oracle import wordpiece_oracle as W
from tests import ckpt_util
from tests.golden_util import assert_update_parity, load
from tests.test_weights import ALBEF_DIMS, albef_params_from, vilt_params_from

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.fixture(scope="module")
def vilt_dir(tmp_path_factory):
    return ckpt_util.write_hf_vilt_checkpoint(str(tmp_path_factory.mktemp("vilt_b32_mlm")), layers=2)


def _dev(b):
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}


def test_vilt_engine_on_loaded_weights_matches_the_reference(vilt_dir, golden_dir):
    from feddat_amd import engine
    rec = load(golden_dir, "g13_vilt_pretrained.npz")
    P, d = vilt_params_from(vilt_dir)
    eng = engine.ViltDatEngine(P, ["art"], DEV, batch=2, res=384, layers=2)
    batch = _dev(O.synthetic_batch(2, 384, 1300))
    for mode in ("gating", "adapter_1"):
        pooled, lg = eng.forward(batch, mode, "art")
        assert float((pooled.cpu() - torch.from_numpy(rec[f"fwd.{mode}.pooled"])).abs().max()) < 2e-2, mode
        assert float((lg.cpu() - torch.from_numpy(rec[f"fwd.{mode}.logits"])).abs().max()) < 3e-2, mode


def test_albef_engine_on_loaded_weights_matches_the_reference(tmp_path, golden_dir):
    from feddat_amd import albef_engine
    rec = load(golden_dir, "g14_albef_pretrained.npz")
    d0 = ALBEF_DIMS
    path = ckpt_util.write_albef_checkpoint(str(tmp_path / "ALBEF.pth"), vit_depth=d0["vit_depth"], enc_layers=d0["enc_layers"],
                                            pre_image=48, vocab=d0["vocab"], max_pos=d0["max_pos"])
    P, d = albef_params_from(path)
    eng = albef_engine.AlbefDatEngine(P, DEV, batch=3, n_answers=6, q_len=12, a_len=5, vit_depth=d0["vit_depth"],
                                      enc_layers=d0["enc_layers"], fusion_layer=d0["fusion_layer"], dec_layers=d0["dec_layers"],
                                      image=d0["image"], vocab=d0["vocab"])
    b0 = _dev(A.synthetic_batch(3, d, 1400, q_len=12, a_len=5, k=[2, 1, 3], ragged=True))
    for mode in ("gating", "adapter_1"):
        loss, logits = eng.forward_train_logits(b0, mode)
        ref_l = float(rec[f"fwd.{mode}.loss"])
        assert abs(float(loss) - ref_l) < 3e-3 * ref_l, (mode, float(loss), ref_l)
        got = logits.float().cpu()[..., :d0["vocab"]]
        assert float((got - torch.from_numpy(rec[f"fwd.{mode}.logits"])).abs().max()) < 6e-2, mode


def test_main_trains_from_a_pretrained_directory(vilt_dir, tmp_path):
    """train.main --pretrained_model_name <HF dir>: the frozen backbone is the file's, the run trains, and a hub NAME (not a
    path) is refused instead of silently training a random backbone."""
    from feddat_amd import train, weights
    argv = ["--encoder_name", "vilt", "--ordered_cl_tasks", "art,gqa", "--comm_rounds", "1", "--batch_size", "2",
            "--num_layers", "2", "--image_size", "224", "--synthetic_steps", "2", "--output_dir", str(tmp_path)]
    model = train.main(argv + ["--pretrained_model_name", vilt_dir])
    ref = weights.convert_vilt_state_dict(weights.read_checkpoint(vilt_dir), 2)
    wq = ref["vilt_encoder.vilt.encoder.layer.1.attention.attention.query.weight"].to(DEV).to(model.engine.op_dtype)
    assert model.engine.op_dtype == torch.float16       # the default operand format (mixed_precision fp16, as the reference)
    assert torch.equal(model.engine.layers[1]["wqkv"][:768], wq)            # the engine's 16-bit operand IS the file's tensor
    assert model.exchange_used in ("feddat_fedavg_allreduce", "none")
    rand = train.main(argv)                                                   # no flag: random weights of the architecture
    assert not torch.equal(rand.engine.layers[1]["wqkv"][:768], wq)
    with pytest.raises(weights.FeddatHipError, match="no such local file"):
        train.main(argv + ["--pretrained_model_name", "dandelin/vilt-b32-mlm"])


# ------------------------------------------------------------------------------------------ raw batches
def _raw_batch(golden_dir, B, seed):
    g = load(golden_dir, "g9_wordpiece.npz")
    vocab = str(g["vocab"]).split("\n")
    texts = [t for t in str(g["texts"]).split("\x1e") if t.isascii() and 0 < len(t) < 200][seed:seed + B]
    assert len(texts) == B
    shapes = [(480, 640), (333, 500), (384, 384), (300, 420), (375, 500), (240, 320)]
    imgs = IO.synthetic_images([shapes[(seed + i) % len(shapes)] for i in range(B)], seed=seed)
    tgt = O.synthetic_batch(B, 32, 50 + seed)["target_scores"]
    return vocab, {"images": imgs, "raw_texts": texts, "target_scores": tgt}


def _oracle_encodings(vocab, raw, frame, max_len=40):
    """The reference's process_inputs on the host: oracle image processor (pinned bit-exact to Pillow / HF, G7) and oracle
    WordPiece (pinned to HF tokenizers, G9), padded to the engine's static frame."""
    rpx, rpm = IO.vilt_image_processor(raw["images"])
    B = len(raw["images"])
    px = torch.zeros(B, 3, *frame)
    pm = torch.zeros(B, *frame, dtype=torch.long)
    px[:, :, :rpx.shape[2], :rpx.shape[3]] = torch.from_numpy(rpx)
    pm[:, :rpm.shape[1], :rpm.shape[2]] = torch.from_numpy(rpm)
    ids, mask, tt = W.encode_batch(raw["raw_texts"], {t: i for i, t in enumerate(vocab)}, max_len, pad_to=max_len)
    return {"pixel_values": px, "pixel_mask": pm, "input_ids": torch.from_numpy(ids).long(),
            "attention_mask": torch.from_numpy(mask).long(), "token_type_ids": torch.from_numpy(tt).long(),
            "target_scores": raw["target_scores"]}


def test_model_takes_images_and_texts(golden_dir):
    """model(task_key, images=[uint8 arrays], texts=[str]) == the oracle on the host-processed batch; the encodings the
    device pipeline produces are bit-identical with the oracle's (images) / HF's (tokens)."""
    from feddat_amd import modeling
    B, frame = 4, (384, 640)
    vocab, raw = _raw_batch(golden_dir, B, 3)
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art"], bias_std=0.02)
    model = modeling.create_vilt_continual_learner_model(P, ["art"], DEV, B, frame, 2, vocab=vocab)
    enc = model.process_inputs(**modeling.convert_batch_to_vilt_input_dict(raw))
    ref = _oracle_encodings(vocab, raw, frame)
    for k in ("pixel_values", "pixel_mask", "input_ids", "attention_mask", "token_type_ids"):
        assert enc[k].dtype == ref[k].dtype and torch.equal(enc[k].cpu(), ref[k]), k
    assert int(ref["attention_mask"].sum(1).min()) < 40 and int(ref["pixel_mask"].sum()) < B * frame[0] * frame[1]
    for mode in ("gating", "adapter_1"):
        if mode == "gating":
            model.activate_gating()
        else:
            model.deactivate_gating()
            model.set_active_adapter(mode)
        pooled, lg = model("art", images=raw["images"], texts=raw["raw_texts"])
        with torch.no_grad():
            rp, rl = O.vilt_forward(P, d, ref, mode, "art")
        assert float((pooled.cpu() - rp).abs().max()) < 3e-2 and float((lg.cpu() - rl).abs().max()) < 3e-2, mode
    # PIL images are accepted like arrays
    try:
        from PIL import Image
        pil = [Image.fromarray(a) for a in raw["images"]]
        p2, l2 = model("art", images=pil, texts=raw["raw_texts"])
        assert torch.equal(l2, lg)
    except ImportError:
        pass
    with pytest.raises(Exception):
        modeling.create_vilt_continual_learner_model(P, ["art"], DEV, B, frame, 2).process_inputs(raw["images"], raw["raw_texts"])


def test_train_step_takes_the_reference_batch_schema(golden_dir):
    """TaskTrainer.train on {"images", "raw_texts", "target_scores"} batches (hipGraph replay + prefetch worker running the
    processor and the tokenizer) vs the oracle client on the host-processed batches: losses and weight updates."""
    from feddat_amd import modeling, train
    import types
    B, frame, steps = 4, (384, 640), 3
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art"], bias_std=0.02)
    P0 = {k: v.clone() for k, v in P.items()}
    raws, vocab = [], None
    for s in range(steps):
        vocab, raw = _raw_batch(golden_dir, B, 5 + 4 * s)
        raws.append(raw)
    model = modeling.create_vilt_continual_learner_model(P, ["art"], DEV, B, frame, 2, vocab=vocab)
    args = types.SimpleNamespace(local_epochs=1, num_epochs=15, lr=1e-4, optimizer_mode="dat", debug=0, hip_graph=True,
                                 prefetch=True)
    tr = train.TaskTrainer(args, "art", raws, raws[:1])
    tr.train(model)
    torch.cuda.synchronize()
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=steps)
    for raw in raws:
        client.train_step(_oracle_encodings(vocab, raw, frame))
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    names = [n for n in sd if "adapter_2" not in n]
    assert_update_parity(names, sd, P, P0, what="raw-batch train")
    # eval on raw batches goes through the same converter (device-side score)
    scores = tr.eval(model)
    assert len(scores) == 3 and all(0.0 <= s <= 100.0 for s in scores)
