"""Streaming row (SURVEY 8a9 + 8f-1): the host-side detector on CPU, the GPU window loop under -m gpu."""
import os

import numpy as np
import pytest

from multilingual_kws_amd.embedding import batch_streaming_analysis as bsa
from multilingual_kws_amd.embedding.single_target_recognize_commands import RecognizeResult, SingleTargetRecognizeCommands
from tests.util_data import tone_clip, write_wav


def _run(scores, thr=0.5, avg_ms=100, sup_ms=500, min_count=4, stride_ms=20):
    rc = SingleTargetRecognizeCommands(["_silence_", "_unknown_", "kw"], avg_ms, thr, sup_ms, min_count, 2)
    el, events = RecognizeResult(), []
    for i, s in enumerate(scores):
        rc.process_latest_result(np.array([1 - s, 0.0, s]), i * stride_ms, el)
        if el.is_new_command:
            events.append((i * stride_ms, el.found_command, round(float(el.score), 6)))
    return events


def test_detector_needs_minimum_count_and_fires_once():
    # 100 ms window at 20 ms hops holds 6 results (times t-100..t); first evaluation once 4 results cover >= 25 ms
    ev = _run([0.9] * 40)
    assert ev[0] == (60, "kw", 0.9)                      # 4th window (t = 60 ms)
    assert [e for e in ev if e[1] == "kw"] == [ev[0]]    # suppressed afterwards: label unchanged, no re-fire


def test_detector_release_and_refire_respect_suppression():
    scores = [0.9] * 10 + [0.0] * 60 + [0.9] * 10
    ev = _run(scores)
    kinds = [(t, c) for t, c, _ in ev]
    assert kinds[0] == (60, "kw")
    # average drops below 0.5 a few windows after t = 200 ms, but the release needs > 500 ms since the last event
    rel = [t for t, c in kinds if c == "_silence_"]
    assert rel and rel[0] == 580                          # first hop with t - 60 > 500
    fire2 = [t for t, c in kinds if c == "kw"][1:]
    assert fire2 and fire2[0] >= rel[0] + 500             # re-fire also waits out the suppression window
    with pytest.raises(ValueError):
        SingleTargetRecognizeCommands(["a", "b", "c"], 100, 0.5, 500, 4, 2).process_latest_result(np.zeros(2), 0, RecognizeResult())


def test_detector_accumulates_in_float64_for_float32_inferences():
    """float32 softmax rows (what model.predict returns) must be averaged in float64 like the reference's np.zeros sum."""
    rc = SingleTargetRecognizeCommands(labels=["_silence_", "_unknown_", "k"], average_window_duration_ms=100, detection_threshold=0.5,
                                       suppression_ms=0, minimum_count=1, target_id=2)
    el = RecognizeResult()
    vals = np.asarray([0.1, 0.7000001, 0.7], dtype=np.float32)
    for i, v in enumerate(vals):
        rc.process_latest_result(np.asarray([0, 0, v], dtype=np.float32), 40 * i, el)
    assert isinstance(el.score, float) and not isinstance(el.score, np.floating)
    assert el.score == sum(float(v) / 3 for v in vals)


def test_detector_average_is_over_the_window():
    # one spike cannot fire: mean over >= 4 windows stays below threshold
    assert [e for e in _run([0.0] * 10 + [1.0] + [0.0] * 10) if e[1] == "kw"] == []
    # three of six above: mean 0.5 is not > 0.5
    assert [e for e in _run([0.0, 1.0] * 20) if e[1] == "kw"] == []


def test_detector_matches_reference_golden_vectors(golden_dir):
    """Step-by-step equality with outputs of the reference's own detector (tests/golden/make_detector_golden.py)."""
    import json
    G = json.load(open(os.path.join(golden_dir, "detector_golden.json")))
    assert len(G["cases"]) == 12
    n_events = 0
    for case in G["cases"]:
        cfg = case["config"]
        rc = SingleTargetRecognizeCommands(["_silence_", "_unknown_", "kw"], cfg["avg"], cfg["thr"], cfg["sup"], cfg["minc"], 2)
        el = RecognizeResult()
        for i, (p, exp) in enumerate(zip(case["probs"], case["outputs"])):
            rc.process_latest_result(np.asarray(p), i * cfg["stride"], el)
            assert el.found_command == exp[0] and bool(el.is_new_command) == exp[2], (cfg, i)
            assert float(el.score) == exp[1], (cfg, i)          # same accumulation order -> bit-equal float64
            n_events += exp[2]
    assert n_events > 50


def test_chunking_matches_the_reference_as_shipped():
    """batch_streaming_analysis.py:72-86 restated by hand for n = 10, max = 4 (and the single-chunk case)."""
    a = np.arange(10)
    assert [c.tolist() for c in bsa.chunk_audio(a, 11)] == [a.tolist()]                      # n < max: one chunk
    got = [c.tolist() for c in bsa.chunk_audio(a, 4)]
    # offsets 0, 4, 8: 0+4 > 10? no -> audio[0:];  4+4 > 10? no -> audio[4:];  8+4 > 10? yes -> audio[8:12]
    assert got == [list(range(10)), list(range(4, 10)), [8, 9]]
    assert [c.tolist() for c in bsa.chunk_audio(a, 10)] == [a.tolist()]                      # n == max: range gives offset 0 only, 0+10 > 10 false
    assert bsa.chunk_audio(a, None)[0] is a


def test_window_offsets_and_flags():
    assert bsa.window_offsets(16000, 16000, 320) == []                       # the reference evaluates NO window here
    assert bsa.window_offsets(16000 + 640, 16000, 320) == [0, 320]
    assert bsa.window_offsets(16000 + 641, 16000, 320) == [0, 320, 640]
    f = bsa.StreamFlags(wav="x.wav", ground_truth="g.txt", target_keyword="kw", detection_thresholds=[0.5])
    assert f.labels() == ["_silence_", "_unknown_", "kw"] and (f.clip_stride_ms, f.average_window_duration_ms, f.suppression_ms, f.minimum_count) == (20, 100, 500, 4)


@pytest.mark.gpu
def test_streaming_inferences_match_per_window_predict(tmp_path):
    """Frame sharing, graph replay and multi-head serving against the package's own per-window predict (HIP vs HIP, bit for bit).  That
    comparison is sound because each leg is oracle-checked on its own -- frame sharing against the C oracle (tests/test_frontend_gpu.py,
    hops 320 / 640 / 1600), the embedding (tests/test_embedding_gpu.py), the heads (tests/test_head_gpu.py) -- and the row does not rest on
    transitivity alone: tests/test_surface.py::test_fifty_keyword_detections_from_one_embedding_pass compares every window of a 50-head
    stream with the CPU oracle chain directly."""
    torch = pytest.importorskip("torch")
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    from multilingual_kws_amd.head import Head
    ms = input_data.standard_microspeech_model_settings(3)
    rng = np.random.default_rng(0)
    pcm = np.concatenate([tone_clip(400 + 300 * k, rng, n=8000) for k in range(7)])       # 3.5 s
    wav = str(tmp_path / "stream.wav")
    write_wav(wav, pcm)
    emb, blob = tl.load_base_model("synthetic", max_batch=256)
    models = [tl.TransferLearnedModel(emb, Head(max_batch=256, seed=s), blob, "synthetic") for s in (1, 2, 3)]
    audio = pcm.astype(np.float32) / 32768
    inf = bsa.streaming_inferences(models, ms, audio)
    offs = bsa.window_offsets(len(pcm), 16000, 320)
    assert all(i.shape == (len(offs), 3) for i in inf) and len(offs) == 125
    # reference semantics: every window = to_micro_spectrogram(slice) -> model.predict
    wins = np.stack([audio[o:o + 16000] for o in offs])
    specs = input_data.to_micro_spectrogram(ms, wins)
    for m, got in zip(models, inf):
        assert np.array_equal(got, m.predict(specs[..., None]))            # bit-identical: frame sharing changes nothing
    assert not np.array_equal(inf[0], inf[1])
    # full batches replay a captured hipGraph, the ragged tail runs eagerly: 125 windows = 3 x 32 + 29, twice (cache hit), vs no graph
    for bw in (32, 32, 16):                # 3 concurrent lanes + tail;  again (cache hit);  4 lanes, then 3, then a tail of 13
        inf_g = bsa.streaming_inferences(models, ms, audio, batch_windows=bw)
        assert all(np.array_equal(a, b) for a, b in zip(inf_g, inf)), bw
    assert all(np.array_equal(a, b) for a, b in zip(bsa.streaming_inferences(models, ms, audio, batch_windows=32, use_graph=False), inf))
    flags = bsa.StreamFlags(wav=wav, ground_truth="", target_keyword="kw", detection_thresholds=[0.3, 0.6])
    results, inferences = bsa.calculate_streaming_accuracy(models[0], ms, [flags])
    assert np.array_equal(inferences, inf[0]) and set(results[0][1]) == {0.3, 0.6}
    found, found_conf = results[0][1][0.3]
    assert all(w == "kw" for w, _ in found) and len(found) == len(found_conf)
    # odd hop (not a multiple of the 320-sample frame step) falls back to explicit windows
    sp = bsa.stream_spectrograms(ms, audio, 16000, 500)
    ref = input_data.to_micro_spectrogram(ms, np.stack([audio[o:o + 16000] for o in bsa.window_offsets(len(pcm), 16000, 500)]))
    assert np.array_equal(sp.cpu().numpy(), ref)
    # max_chunk_length_sec (reference :72-86, as shipped): 3.5 s with 2 s chunks -> chunks audio[0:], then audio[32000:64000]
    flags2 = bsa.StreamFlags(wav=wav, ground_truth="", target_keyword="kw", detection_thresholds=[0.3], max_chunk_length_sec=2)
    _, inf_chunked = bsa.calculate_streaming_accuracy(models[0], ms, [flags2])
    tail_offs = bsa.window_offsets(len(pcm) - 32000, 16000, 320)
    assert inf_chunked.shape == (len(offs) + len(tail_offs), 3)
    assert np.array_equal(inf_chunked[:len(offs)], inf[0])
    tail = np.stack([audio[32000 + o:32000 + o + 16000] for o in tail_offs])
    assert np.array_equal(inf_chunked[len(offs):], models[0].predict(input_data.to_micro_spectrogram(ms, tail)[..., None]))


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 4])
def test_streaming_session_graph_replay_equals_eager_predict(batch):
    """StreamingSession: the live window loop as one hipGraph replay per window.  Every replay must equal the
    reference semantics -- to_micro_spectrogram(window) -> model.predict -- bit for bit, for every keyword head."""
    torch = pytest.importorskip("torch")
    from multilingual_kws_amd import synth
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    from multilingual_kws_amd.head import Head
    ms = input_data.standard_microspeech_model_settings(3)
    emb, blob = tl.load_base_model("synthetic", max_batch=batch)
    models = [tl.TransferLearnedModel(emb, Head(max_batch=batch, seed=s), blob, "synthetic") for s in range(5)]
    sess = bsa.StreamingSession(models, ms, batch=batch)
    assert sess.graph is not None
    if batch == 1 and emb.get_option("fuse_cluster"):
        assert emb.get_option("fuse_cluster_chain") == 1        # round 6: a live window's blocks 4b .. 7a replay as ONE cluster-chain launch
    eager = bsa.StreamingSession(models, ms, batch=batch, use_graph=False)
    clips = synth.clips_float32(3 * batch)
    for k in (0, 1, 2, 1):
        a = clips[k * batch:(k + 1) * batch]
        got = sess.infer(a if k else torch.from_numpy(a).cuda()).clone()            # numpy and CUDA inputs
        torch.cuda.synchronize()
        specs = input_data.to_micro_spectrogram(ms, a)
        for m, g in zip(models, got):
            assert np.array_equal(g.cpu().numpy(), m.predict(specs[..., None]))
        assert torch.equal(eager.infer(a), got)
    with pytest.raises(ValueError):
        bsa.StreamingSession(models, ms, batch=batch + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 64])
def test_graph_serving_recovers_from_a_failed_exchange(batch, tmp_path):
    """A failed pair / cluster exchange poisons the forward that ran it (all-NaN embeddings) and is reported by the NEXT eager call --
    but a captured hipGraph never makes an eager call.  StreamingSession / streaming_inferences therefore poll the handle's
    "exchange_error" word: the session re-captures before its next replay (one poisoned window, never NaN for ever), the offline
    stream is repeated on the healed handles and returns clean results."""
    torch = pytest.importorskip("torch")
    import warnings
    from multilingual_kws_amd import synth
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    from multilingual_kws_amd.head import Head
    ms = input_data.standard_microspeech_model_settings(3)
    emb, blob = tl.load_base_model("synthetic", max_batch=batch)
    if emb.get_option("fuse_pair") != 1:
        pytest.skip("the exchange kernels are not in this handle's plan on this device")
    models = [tl.TransferLearnedModel(emb, Head(max_batch=batch, seed=s), blob, "synthetic") for s in range(3)]
    sess = bsa.StreamingSession(models, ms, batch=batch)
    clips = synth.clips_float32(2 * batch)
    good = [sess.infer(clips[k * batch:(k + 1) * batch]).clone() for k in range(2)]
    assert all(torch.isfinite(g).all() for g in good) and sess.recaptures == 0
    emb.set_option("inject_exchange_error", 1)                  # as if the previous replay's exchange had failed
    assert emb.get_option("exchange_error") != 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        again = [sess.infer(clips[k * batch:(k + 1) * batch]).clone() for k in range(2)]
    torch.cuda.synchronize()
    assert sess.recaptures == 1 and emb.get_option("exchange_error") == 0 and emb.get_option("fuse_pair") == 0 and emb.get_option("pair_degraded") == 1
    for g, a in zip(good, again):                               # healed plan = other kernels: equal up to fp32 rounding, labels exact
        assert torch.isfinite(a).all() and torch.allclose(a, g, rtol=1e-4, atol=1e-6) and torch.equal(a.argmax(-1), g.argmax(-1))
    # offline stream: the failure is met INSIDE the run (graph replays give no return code); the result must still be clean
    emb2, _ = tl.load_base_model("synthetic", max_batch=batch)
    models2 = [tl.TransferLearnedModel(emb2, m.head, blob, "synthetic") for m in models]
    rng = np.random.default_rng(3)
    pcm = np.concatenate([tone_clip(500 + 200 * k, rng, n=8000) for k in range(8)])       # 4 s: 150 windows
    audio = pcm.astype(np.float32) / 32768
    ref = bsa.streaming_inferences(models2, ms, audio, batch_windows=min(batch, 32))
    for e in emb2.replicas(1 if emb2.get_option("fuse_cluster") else bsa.SERVING_LANES):
        e.set_option("fuse_pair", 1)
    orig_run, state = bsa._BatchGraph.run_all, {"calls": 0}

    def run_then_fail(self, *args):                             # the exchange "fails" after the first chunk's replays
        out = orig_run(self, *args)
        state["calls"] += 1
        if state["calls"] == 1:
            self.keep[0][0].set_option("inject_exchange_error", 1)   # (lane 0: the caller's handle, or its first serving replica)
        return out
    bsa._BatchGraph.run_all = run_then_fail
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = bsa.streaming_inferences(models2, ms, audio, batch_windows=min(batch, 32))
    finally:
        bsa._BatchGraph.run_all = orig_run
    assert emb2.get_option("exchange_error") == 0 and sum(e.get_option("pair_degraded") for e in [emb2] + list(emb2._replicas)) == 1
    assert any("repeating the stream" in str(x.message) for x in w)
    for r, g in zip(ref, got):
        assert np.isfinite(g).all() and np.allclose(g, r, rtol=1e-4, atol=1e-6) and np.array_equal(g.argmax(1), r.argmax(1))


@pytest.mark.gpu
def test_offline_stream_raises_when_the_repeat_fails_too(tmp_path):
    """streaming_inferences repeats a stream once after a failed in-graph exchange -- with EVERY serving-lane replica taken off the exchange
    kernels first -- and must not hand back poisoned probabilities if the repeat reports a failure as well: MkwsError(MKWS_ERR_EXCHANGE)."""
    torch = pytest.importorskip("torch")
    import warnings
    from multilingual_kws_amd import _lib
    from multilingual_kws_amd.embedding import input_data, transfer_learning as tl
    from multilingual_kws_amd.head import Head
    ms = input_data.standard_microspeech_model_settings(3)
    emb, blob = tl.load_base_model("synthetic", max_batch=64)
    if emb.get_option("fuse_pair") != 1:
        pytest.skip("the exchange kernels are not in this handle's plan on this device")
    models = [tl.TransferLearnedModel(emb, Head(max_batch=64, seed=s), blob, "synthetic") for s in range(2)]
    rng = np.random.default_rng(4)
    pcm = np.concatenate([tone_clip(500 + 200 * k, rng, n=8000) for k in range(8)])
    audio = pcm.astype(np.float32) / 32768
    ref = bsa.streaming_inferences(models, ms, audio, batch_windows=32)
    orig_run = bsa._BatchGraph.run_all

    def always_fail(self, *args):                               # every chunk's replays leave the error word set, as a failed exchange would
        out = orig_run(self, *args)
        self.keep[0][0].set_option("inject_exchange_error", 1)
        return out
    bsa._BatchGraph.run_all = always_fail
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            with pytest.raises(_lib.MkwsError) as ei:
                bsa.streaming_inferences(models, ms, audio, batch_windows=32)
        assert ei.value.code == _lib.MKWS_ERR_EXCHANGE
    finally:
        bsa._BatchGraph.run_all = orig_run
    # every handle of the stream left the exchange kernels before the repeat; the stream runs clean again afterwards
    reps = [emb] + list(emb._replicas)
    assert all(e.get_option("fuse_pair") == 0 and e.get_option("fuse_cluster") == 0 for e in reps)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        emb.forward(torch.zeros((1, 49, 40), device=emb.device))          # (clears the injected word: the wrapper repeats the call that reports it)
    again = bsa.streaming_inferences(models, ms, audio, batch_windows=32)
    for r, g in zip(ref, again):
        assert np.isfinite(g).all() and np.allclose(g, r, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_serving_lanes_run_side_by_side_and_agree_with_the_eager_path(monkeypatch):
    """serve_spectrograms with several lanes: every full batch on a replica handle with the workgroup shapes of lanes x batch clips (option
    "plan_batch"), one captured graph per lane on its own stream, ONE join per call, the ragged tail launched eagerly beside the lanes.  The
    result must be the eager path's up to the round-off between workgroup shapes (labels exact), equal from call to call, and -- with one lane,
    i.e. the caller's own handle and plan -- equal to the eager path bit for bit.  The caller's handle keeps its own plan."""
    torch = pytest.importorskip("torch")
    from multilingual_kws_amd.embedding import transfer_learning as tl
    from multilingual_kws_amd.head import Head
    emb, _ = tl.load_base_model("synthetic", max_batch=256)
    heads = [Head(max_batch=256, seed=40 + s) for s in range(3)]
    g = torch.Generator(device="cpu").manual_seed(11)
    specs = (torch.rand((5 * 256 + 77, 49, 40), generator=g) * 26).cuda()
    ref = bsa.serve_spectrograms(emb, heads, specs, 256, use_graph=False)
    assert tuple(ref.shape) == (3, 5 * 256 + 77, 3)
    monkeypatch.setattr(bsa, "SERVING_LANES", 4)
    got = bsa.serve_spectrograms(emb, heads, specs, 256).clone()
    again = bsa.serve_spectrograms(emb, heads, specs, 256)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and torch.equal(got, again)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-6) and torch.equal(got.argmax(-1), ref.argmax(-1))
    assert torch.equal(got[:, 5 * 256:], ref[:, 5 * 256:])                    # the ragged tail ran on the caller's handle
    assert emb.get_option("plan_batch") == 256 and len(emb._replicas) == 4
    assert all(r.get_option("plan_batch") == 1024 and r.get_option("block_tiles") == 3 for r in emb._replicas)
    monkeypatch.setattr(bsa, "SERVING_LANES", 1)
    one = bsa.serve_spectrograms(emb, heads, specs, 256)
    torch.cuda.synchronize()
    assert torch.equal(one, ref)
    with pytest.raises(Exception):
        emb.set_option("plan_batch", 128)                                     # below max_batch
    # the lanes' streams were measured to overlap pairwise (one hardware queue each) and are handed out again; the lane count respects the
    # paired kernels' budget: lanes x 2 x ceil(batch / 8) workgroups that hold their CU while they wait must fit the chip
    from multilingual_kws_amd import streams
    got4 = streams.concurrent_streams(4, emb.device)
    assert 1 <= len(got4) <= 4 and len({s_.cuda_stream for s_ in got4}) == len(got4)
    assert [s_.cuda_stream for s_ in streams.concurrent_streams(2, emb.device)] == [s_.cuda_stream for s_ in got4[:2]]
    cus = torch.cuda.get_device_properties(emb.device).multi_processor_count
    assert bsa.lane_budget(256, emb.device) == max(1, cus // 64) and bsa.lane_budget(1024, emb.device) == max(1, cus // 256)
