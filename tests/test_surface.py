"""The rest of the reference's Python surface for the streaming row: the `multilingual_kws.*` import path, tpr_fpr,
StreamTarget / eval_stream_test, run.inference and the multi-keyword detections dict.  CPU tests stub the GPU window loop; the -m gpu
test runs 50 keyword heads on one shared embedding pass and compares every window with the CPU oracle chain directly."""
import json
import os
import pickle

import numpy as np
import pytest

from tests.conftest import ROOT
from tests.util_data import tone_clip, write_wav


def test_reference_import_path_resolves_to_the_build():
    """run.py:15-18 of the reference, verbatim."""
    from multilingual_kws.embedding import input_data
    from multilingual_kws.embedding import batch_streaming_analysis as sa
    from multilingual_kws.embedding import transfer_learning
    from multilingual_kws.embedding.tpr_fpr import tpr_fpr, get_groundtruth
    import multilingual_kws
    import multilingual_kws.embedding.input_data as by_path
    import multilingual_kws_amd.embedding.batch_streaming_analysis as amd_sa
    import multilingual_kws_amd.embedding.input_data as amd_input_data
    import multilingual_kws_amd.embedding.transfer_learning as amd_tl
    assert os.path.dirname(os.path.abspath(multilingual_kws.__file__)) == os.path.join(ROOT, "multilingual_kws")
    assert input_data is amd_input_data is by_path and sa is amd_sa and transfer_learning is amd_tl      # the same module objects
    from multilingual_kws import run
    import multilingual_kws.run as run2
    import multilingual_kws_amd.run as amd_run
    assert run is run2 is amd_run and callable(run.inference)
    for name in ("StreamFlags", "StreamTarget", "eval_stream_test", "calculate_streaming_accuracy"):
        assert hasattr(sa, name)
    assert callable(tpr_fpr) and callable(get_groundtruth) and callable(transfer_learning.transfer_learn)
    assert input_data.standard_microspeech_model_settings(3)["fingerprint_size"] == 1960
    import dataclasses
    assert [f.name for f in dataclasses.fields(sa.StreamTarget)] == ["target_lang", "target_word", "model_path", "stream_flags",
                                                                      "destination_result_pkl", "destination_result_inferences"]


def test_tpr_fpr_matches_reference_golden_vectors(golden_dir, capsys):
    """Outputs of the reference's own tpr_fpr.py on seeded inputs (tests/golden/make_tpr_fpr_golden.py), incl. unsorted detection lists
    (its scans give up at the first entry past the window) and get_groundtruth's return inside the loop over targets."""
    from multilingual_kws_amd.embedding import tpr_fpr as T
    G = json.load(open(os.path.join(golden_dir, "tpr_fpr_golden.json")))
    assert len(G["get_groundtruth"]) == 12 and len(G["tpr_fpr"]) == 12
    kinds = set()
    for c in G["get_groundtruth"]:
        gt = [tuple(x) for x in c["groundtruth"]]
        got = T.get_groundtruth(c["found"], c["targets"], gt) if c["tol"] is None else T.get_groundtruth(c["found"], c["targets"], gt, c["tol"])
        assert got == c["out"]
        kinds |= {d["groundtruth"] for d in got}
        assert {d["keyword"] for d in got} <= {c["targets"][0]}             # as shipped: the first target only
        if len(c["targets"]) > 1:
            more = T.get_groundtruth(c["found"], c["targets"], gt, c["tol"] or 1500, first_target_only=False)
            assert more[:len(got)] == got and len(more) >= len(got)
    assert kinds == {"tp", "fp", "fn"}
    for c in G["tpr_fpr"]:
        assert T.tpr_fpr(c["keyword"], c["thresh"], c["found"], c["gt_times"], c["duration_s"], c["tol"], c["nontarget"]) == c["out"]
    capsys.readouterr()


def _stream_wav(tmp_path, seconds=3.0, seed=0):
    rng = np.random.default_rng(seed)
    pcm = np.concatenate([tone_clip(400 + 300 * k, rng, n=8000) for k in range(int(seconds * 2))])
    wav = str(tmp_path / "stream.wav")
    write_wav(wav, pcm)
    return wav, pcm


def _bursty_inferences(n, seed, centres):
    """[n, 3] softmax-like rows whose target column rises around the given windows."""
    rng = np.random.default_rng(seed)
    tgt = np.full(n, 0.02)
    for c in centres:
        tgt[max(0, c - 8):c + 8] = 0.97
    other = rng.uniform(0, 1, n) * (1 - tgt)
    return np.stack([1 - tgt - other, other, tgt], axis=1).astype(np.float32)


def test_eval_stream_test_contract_on_stored_inferences(tmp_path, capsys):
    """Reference :198-241 with the window loop out of the picture (stored inferences): result dict, pickle, early return."""
    from multilingual_kws_amd.embedding import batch_streaming_analysis as sa
    wav, pcm = _stream_wav(tmp_path)
    n = len(sa.window_offsets(len(pcm), 16000, 320))
    inf = _bursty_inferences(n, 1, [30, 80])
    npy, pkl = str(tmp_path / "raw_inferences.npy"), str(tmp_path / "stream_results.pkl")
    np.save(npy, inf)
    flags = sa.StreamFlags(wav=wav, ground_truth="", target_keyword="mask", detection_thresholds=[0.5, 0.9])
    st = sa.StreamTarget(target_lang="en", target_word="mask", model_path="unused", stream_flags=[flags],
                         destination_result_pkl=pkl, destination_result_inferences=npy)
    res = sa.eval_stream_test(st, live_model=object())
    assert list(res) == ["mask"] and len(res["mask"]) == 1
    got_flags, by_thresh = res["mask"][0]
    assert got_flags == flags and list(by_thresh) == [0.5, 0.9]
    found, found_conf = by_thresh[0.9]
    assert [w for w, _ in found] == ["mask", "mask"] and [t for _, t in found] == [t for _, t, _ in found_conf]
    assert all(0.9 < c <= 1.0 for _, _, c in found_conf)
    # what run.py:113 reads: results[keyword][0][1][detection_threshold][1]
    assert res["mask"][0][1][0.9][1] == found_conf
    assert pickle.load(open(pkl, "rb")) == res
    assert np.array_equal(np.load(npy), inf)                                 # re-used, not rewritten
    assert sa.eval_stream_test(st, live_model=object()) is None              # "results already present"
    assert "results already present" in capsys.readouterr().out
    # no destinations: nothing written
    st2 = sa.StreamTarget("en", "mask", "unused", [flags])
    os.remove(pkl)
    real = sa.streaming_inferences
    sa.streaming_inferences = lambda model, ms, audio, *a, **k: inf
    try:
        assert sa.eval_stream_test(st2, live_model=object()) == res and not os.path.exists(pkl)
        # inferences computed (stubbed) and saved when only the .npy destination is new
        npy2 = str(tmp_path / "new.npy")
        sa.eval_stream_test(sa.StreamTarget("en", "mask", "unused", [flags], None, npy2), live_model=object())
        assert np.array_equal(np.load(npy2), inf)
    finally:
        sa.streaming_inferences = real
    capsys.readouterr()


def test_multi_keyword_detections_dict_of_run_py(tmp_path, capsys):
    """run.py:89-152: per-keyword detections merged, sorted by time, wrapped as dict(keywords, detections, min_threshold); "ng" without
    a ground-truth file, tp / fp / fn (first keyword, as shipped) with one; JSON written.  The GPU window loop is stubbed."""
    from multilingual_kws_amd import run
    from multilingual_kws_amd.embedding import batch_streaming_analysis as sa
    wav, pcm = _stream_wav(tmp_path, seconds=4.0)
    n = len(sa.window_offsets(len(pcm), 16000, 320))
    per_kw = {"alpha": _bursty_inferences(n, 1, [40, 120]), "beta": _bursty_inferences(n, 2, [20, 90]), "gamma": _bursty_inferences(n, 3, [])}

    class Model:            # what load_models_shared returns, as far as the merge logic looks
        def __init__(self, kw, emb):
            self.kw, self.embedding = kw, emb

        def predict(self, x):
            raise AssertionError("not used")
    shared = object()
    models = [Model(k, shared) for k in per_kw]
    calls, real = [], sa.streaming_inferences

    def fake(models_, ms, audio, sample_rate, clip_ms, stride_ms, max_chunk_length_sec=None, **k):
        calls.append((len(models_), max_chunk_length_sec))
        return [per_kw[m.kw] for m in models_]
    sa.streaming_inferences = fake
    try:
        out_json = str(tmp_path / "detections.json")
        det = run.inference(list(per_kw), models, wav, detection_threshold=0.9, inference_chunk_len_seconds=600, write_detections=out_json)
        assert calls == [(3, 600)]                                          # ONE pass for the three keywords
        assert det["keywords"] == ["alpha", "beta", "gamma"] and det["min_threshold"] == 0.9
        d = det["detections"]
        assert [x["keyword"] for x in d] == ["beta", "alpha", "beta", "alpha"]
        assert [x["time_ms"] for x in d] == sorted(x["time_ms"] for x in d)
        assert all(set(x) == {"keyword", "time_ms", "confidence", "groundtruth"} and x["groundtruth"] == "ng" and x["confidence"] > 0.9 for x in d)
        assert json.load(open(out_json)) == det
        # per keyword = what eval_stream_test yields for the same flags (run.py:95-113)
        for kw in per_kw:
            flags = sa.StreamFlags(wav=wav, ground_truth=None, target_keyword=kw, detection_thresholds=[0.9], average_window_duration_ms=100,
                                   suppression_ms=500, time_tolerance_ms=750, max_chunk_length_sec=600)
            sa.streaming_inferences = lambda m, *a, **k: per_kw[kw]
            single = sa.eval_stream_test(sa.StreamTarget("unspecified_language", kw, "unused", [flags]), live_model=object())
            assert [[x["keyword"], x["time_ms"], x["confidence"]] for x in d if x["keyword"] == kw] == single[kw][0][1][0.9][1]
        sa.streaming_inferences = fake
        # ground-truth file: rows keyword,time_ms
        gt = str(tmp_path / "gt.txt")
        t_alpha = [x["time_ms"] for x in d if x["keyword"] == "alpha"]
        open(gt, "w").write(f"alpha,{t_alpha[0] - 100}\nalpha,9999999\nbeta,1\n")
        det2 = sa.multi_keyword_detections(list(per_kw), models, wav, 0.9, 600, groundtruth=gt)
        assert [(x["keyword"], x["groundtruth"]) for x in det2["detections"]] == [("alpha", "fn"), ("alpha", "tp"), ("alpha", "fp")]
        # a single keyword given as a string (run.py:59-61) and the argument checks
        det3 = run.inference("alpha", [models[0]], wav)
        assert det3["keywords"] == ["alpha"] and len(det3["detections"]) == 2
        with pytest.raises(AssertionError):
            run.inference(["alpha", "beta"], [models[0]], wav)
        with pytest.raises(AssertionError):
            run.inference(["alpha"], [models[0]], str(tmp_path / "missing.wav"))
        with pytest.raises(NotImplementedError):
            run.inference(["alpha"], [models[0]], wav, visualizer=True)
    finally:
        sa.streaming_inferences = real
    capsys.readouterr()


@pytest.mark.gpu
def test_fifty_keyword_detections_from_one_embedding_pass(tmp_path, capsys):
    """50 saved few-shot models -> load_models_shared -> ONE embedding handle; run.inference's detections = the per-keyword
    eval_stream_test detections merged and sorted; and -- directly, not through the package's own predict -- every window of the stream
    against the CPU oracle chain (C micro-frontend on the window's samples -> PyTorch-CPU EfficientNet -> numpy head)."""
    torch = pytest.importorskip("torch")
    from multilingual_kws_amd import run, weights
    from multilingual_kws_amd.embedding import batch_streaming_analysis as sa, input_data, transfer_learning as tl
    from multilingual_kws_amd.head import Head
    from oracle import head_oracle as ho
    from oracle.efficientnet_oracle import EmbeddingOracle
    from oracle.frontend_oracle import FrontendOracle
    ms = input_data.standard_microspeech_model_settings(3)
    wav, pcm = _stream_wav(tmp_path, seconds=8.0, seed=5)
    audio = pcm.astype(np.float32) / 32768
    emb, blob = tl.load_base_model("synthetic", max_batch=256)
    keywords = [f"kw{k:02d}" for k in range(50)]
    paths = []
    for k, kw in enumerate(keywords):          # heads with a strong target bias so that some keywords fire on this stream
        p = ho.glorot_uniform_params(seed=2000 + k)
        p[-1] += 0.5 + 0.1 * (k % 7)
        m = tl.TransferLearnedModel(emb, Head(max_batch=256, params=p), blob, "synthetic")
        paths.append(str(tmp_path / f"model_{kw}"))
        m.save(paths[-1])
    models = tl.load_models_shared(paths, max_batch=256)
    assert len({id(m.embedding) for m in models}) == 1 and models[0].embedding is not emb
    thr = 0.5
    det = run.inference(keywords, ",".join(paths), wav, detection_threshold=thr, write_detections=str(tmp_path / "d.json"))
    d = det["detections"]
    assert det["keywords"] == keywords and det["min_threshold"] == thr and len(d) > 10 and len({x["keyword"] for x in d}) > 3
    assert [x["time_ms"] for x in d] == sorted(x["time_ms"] for x in d) and all(x["groundtruth"] == "ng" for x in d)
    # the same call on the live models (run.inference's own load plans its handle for 1024 windows, these are planned for 256: equal up
    # to fp32 round-off in the confidences, same detections)
    live = run.inference(keywords, models, wav, detection_threshold=thr)["detections"]
    assert [(x["keyword"], x["time_ms"]) for x in live] == [(x["keyword"], x["time_ms"]) for x in d]
    assert max(abs(x["confidence"] - y["confidence"]) for x, y in zip(live, d)) < 1e-5
    merged = []
    for kw, m in zip(keywords[:6], models[:6]):          # the reference's route: one eval_stream_test per keyword (same handle: bit for bit)
        flags = sa.StreamFlags(wav=wav, ground_truth=None, target_keyword=kw, detection_thresholds=[thr])
        res = sa.eval_stream_test(sa.StreamTarget("unspecified_language", kw, "unused", [flags]), live_model=m)
        mine = [[x["keyword"], x["time_ms"], x["confidence"]] for x in live if x["keyword"] == kw]
        assert mine == res[kw][0][1][thr][1]
        merged += mine
    assert merged
    # direct oracle comparison of EVERY window of the stream (frame sharing, graph replay and multi-head launch included), four of the 50 heads
    inf = sa.streaming_inferences(models, ms, audio)
    offs = sa.window_offsets(len(pcm), 16000, 320)
    pick = list(range(len(offs)))                        # every window of the stream (350): the oracle chain takes a second
    wins = np.stack([audio[offs[i]:offs[i] + 16000] for i in pick])
    ref_spec = FrontendOracle().run_batch_f32(wins)
    ref_emb = EmbeddingOracle(blob).forward(ref_spec).numpy()
    for k in (0, 7, 23, 49):
        p = models[k].head.get_params()
        ref, _ = ho.forward(p, ref_emb)
        got = inf[k][pick]
        assert np.abs(got - ref).max() < 1e-4 and np.array_equal(got.argmax(1), ref.argmax(1)), k
    capsys.readouterr()
