"""Drop-in for the streaming-inference core of multilingual_kws/embedding/batch_streaming_analysis.py.

The reference (:99-117) slices a long recording into 1 s windows every 20 ms, calls the micro-frontend op
on each window in a Python loop and runs one full model per keyword.  Here one call produces every
window's features on the GPU with per-frame FFT / filterbank work shared across the 49x-overlapping
windows (mkws_frontend_stream_f32; bit-identical to per-window calls), the embedding is computed once and
any number of few-shot heads are applied to it.  The detector (SingleTargetRecognizeCommands) stays on the
host.  StreamTarget / eval_stream_test (:188-241) are the per-keyword entry points run.py drives; multi_keyword_detections is
their multi-keyword form on ONE shared embedding pass (run.py:89-152 runs one full model per keyword)."""
import os
import pickle
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import input_data
from ..streams import concurrent_streams
from .single_target_recognize_commands import RecognizeResult, SingleTargetRecognizeCommands


@dataclass(frozen=True)
class StreamFlags:
    wav: os.PathLike
    ground_truth: os.PathLike
    target_keyword: str
    detection_thresholds: List[float]
    clip_duration_ms: int = 1000
    clip_stride_ms: int = 20
    average_window_duration_ms: int = 100
    suppression_ms: int = 500
    time_tolerance_ms: int = 750
    minimum_count: int = 4
    max_chunk_length_sec: int = 1200

    def labels(self) -> List[str]:
        return [input_data.SILENCE_LABEL, input_data.UNKNOWN_WORD_LABEL, self.target_keyword]


def window_offsets(data_samples, clip_duration_samples, clip_stride_samples):
    """Start sample of every window the reference evaluates: range(0, data_samples - clip, stride)."""
    return list(range(0, data_samples - clip_duration_samples, clip_stride_samples))


def chunk_audio(audio, max_chunk_samples):
    """The reference's chunking of a long recording (:72-86), AS SHIPPED: recordings shorter than max_chunk_samples are
    one chunk; otherwise, for offset in range(0, n, max_chunk_samples), a chunk that would run past the end is cut to
    [offset, offset + max) and every OTHER chunk is the whole remainder audio[offset:] (the condition is inverted in the
    reference, so all chunks but the last overlap everything after them).  Each chunk is windowed on its own and the
    inferences are concatenated, which is what callers of the reference get back."""
    n = audio.shape[0]
    if max_chunk_samples is None or n < max_chunk_samples:
        return [audio]
    chunks = []
    for offset in range(0, n, int(max_chunk_samples)):
        if offset + max_chunk_samples > n:
            chunks.append(audio[offset:offset + int(max_chunk_samples)])
        else:
            chunks.append(audio[offset:])
    return chunks


def stream_spectrograms(model_settings, audio, clip_duration_samples, clip_stride_samples):
    """float32 audio [n] -> CUDA tensor [num_windows, frames, channels], windows as window_offsets()."""
    import torch
    audio_t = torch.as_tensor(np.asarray(audio, dtype=np.float32)) if not torch.is_tensor(audio) else audio
    audio_t = audio_t.cuda().contiguous()
    n = audio_t.shape[0]
    nwin = len(window_offsets(n, clip_duration_samples, clip_stride_samples))
    frames, chans = model_settings["spectrogram_length"], model_settings["fingerprint_width"]
    if nwin <= 0:
        return torch.empty((0, frames, chans), dtype=torch.float32, device=audio_t.device)
    fe = input_data._frontend_for(model_settings, n)
    if clip_stride_samples % model_settings["window_stride_samples"] != 0:
        # hop not a multiple of the frame step: no frame sharing possible, fall back to explicit windows
        idx = (torch.arange(clip_duration_samples, device=audio_t.device)[None]
               + torch.arange(nwin, device=audio_t.device)[:, None] * clip_stride_samples)
        return fe.forward(audio_t[idx])
    return fe.stream(audio_t, clip_duration_samples, clip_stride_samples)[:nwin]


class _BatchGraph:
    """embedding.forward + every head over `lanes` FULL batches of `batch` spectrograms, one captured hipGraph PER LANE, each replayed on its
    own HIP stream (static inputs / outputs); cached per (embedding handle, head handles, batch, lanes).

    Why lanes: at 256 windows every launch of the embedding is latency-bound -- one clip per workgroup, each workgroup streaming the whole
    block's weights -- and a 256-clip batch costs about half of what a 1024-clip batch costs.  The batches of a stream are independent: `lanes`
    of them run side by side, every lane on a replica handle (= its own workspace; the heads are read-only and shared) that runs the PLAN of
    lanes x batch clips (4-clip workgroups, 8-clip pairs: a quarter of the chip per launch at four lanes; EmbeddingModel.serving_lanes).
    Round 6: the lanes used to be the branches of ONE forked hipGraph, which this HIP runtime replays one branch after the other (649 k clips/s
    at 4 x 256 where four graphs on four streams give 1.03 M; profiles/r06_notes.md section 8).  One lane = the caller's own handle and plan,
    the same launches as an eager call, bit for bit."""
    _cache = {}

    def __init__(self, embedding, heads, batch, lanes=1):
        import torch
        dev = embedding.device
        ems = embedding.serving_lanes(lanes, max(lanes, SERVING_LANES) * batch) if lanes > 1 else [embedding]
        self.keep = (ems, list(heads))                               # the graphs hold raw handles: keep their owners alive
        self.owner = embedding
        self.lanes = lanes
        self.heals = 0                                               # re-captures after a failed exchange (run)
        self.generation = (embedding.generation, tuple(h.generation for h in heads))
        self.specs = [torch.zeros((batch, 49, 40), dtype=torch.float32, device=dev) for _ in range(lanes)]
        # one stream per lane, each on a hardware queue of its own (measured, not assumed: streams.py); a single lane runs on the caller's stream
        self.side = concurrent_streams(lanes, dev) if lanes > 1 else [None]
        assert len(self.side) == lanes, "serve_spectrograms asks for no more lanes than concurrent_streams() finds"
        self._capture()

    def _capture(self):
        import torch
        from ..head import Head
        ems, heads = self.keep
        lanes, dev, side = self.lanes, ems[0].device, self.side

        def chain(i):
            return Head.forward_many(heads, ems[i].forward(self.specs[i]))
        warm = [torch.cuda.Stream(device=dev)] if lanes == 1 else side
        for i in range(lanes):                                       # eager warm-up on side streams (lazy init outside the capture); a
            warm[i].wait_stream(torch.cuda.current_stream(dev))      # handle with a recorded exchange failure is healed here (the wrapper
            with torch.cuda.stream(warm[i]):                         # repeats the call that returns MKWS_ERR_EXCHANGE)
                chain(i)
            torch.cuda.current_stream(dev).wait_stream(warm[i])
        self.graphs, self.probs = [], []
        for i in range(lanes):
            g = torch.cuda.CUDAGraph()
            if lanes == 1:
                with torch.cuda.graph(g):
                    self.probs.append(chain(i))
            else:
                with torch.cuda.graph(g, stream=side[i]):
                    self.probs.append(chain(i))
            self.graphs.append(g)

    @classmethod
    def get(cls, embedding, heads, batch, lanes=1):
        # keyed on the handles AND their generation counters: a handle closed and re-created at the same address must not hit a graph
        # that still points into the freed one
        key = (id(embedding), embedding.h.value, tuple(h.h.value for h in heads), int(batch), int(lanes))
        gen = (embedding.generation, tuple(h.generation for h in heads))
        g = cls._cache.get(key)
        if g is not None and g.generation != gen:
            g = None
        if g is None:
            if len(cls._cache) >= 8:
                cls._cache.clear()
            g = cls._cache[key] = cls(embedding, heads, batch, lanes)
        return g

    @classmethod
    def forget(cls, obj):
        """Drop every cached graph that captured `obj` (an EmbeddingModel or Head being closed)."""
        for k in [k for k, g in cls._cache.items() if obj is g.owner or any(obj is e for e in g.keep[0]) or any(obj is h for h in g.keep[1])]:
            del cls._cache[k]

    def exchange_failed(self):
        return any(e.get_option("exchange_error") for e in self.keep[0])

    def run(self, parts):
        """parts: `lanes` tensors [batch,49,40] -> list of [n_heads, batch, 3] (views of the static outputs).  Asynchronous: the results are
        ordered behind the caller's current stream, like any other launch on it."""
        import torch
        if self.exchange_failed():
            # an earlier replay ran a failed pair / cluster exchange (its results were NaN): the captured launches would poison every
            # later batch too.  Heal the handles (eager pass) and capture again -- the new graphs hold the single-workgroup kernels
            self._capture()
            self.heals += 1
        if self.lanes == 1:
            self.specs[0].copy_(parts[0])
            self.graphs[0].replay()
            return self.probs
        main = torch.cuda.current_stream(self.owner.device)
        for i in range(self.lanes):                                  # fork: lane i starts where the caller's stream is now (its input is ready there) ...
            self.side[i].wait_stream(main)
            with torch.cuda.stream(self.side[i]):
                self.specs[i].copy_(parts[i])
                self.graphs[i].replay()
        for i in range(self.lanes):                                  # ... and the caller's stream continues behind every lane
            main.wait_stream(self.side[i])
        return self.probs

    def run_all(self, specs, nfull, tail=None):
        """The first `nfull` FULL batches of specs [windows, 49, 40] -> [n_heads, nfull * batch, 3] on the caller's stream.  Batch j runs on lane
        j % lanes; a lane works through its batches back to back on its own stream (copy in, replay, copy out) and the caller's stream joins the
        lanes ONCE, after `tail()` (the eager pass over a ragged last batch, which so runs beside the lanes).  A join after every round of
        `lanes` batches costs a third of the throughput: the lanes drift apart and every round then waits for its slowest (0.99 -> 1.34 ms per
        4 x 256 clips, profiles/r06_notes.md section 8)."""
        import torch
        if self.exchange_failed():
            self._capture()
            self.heals += 1
        bw = self.specs[0].shape[0]
        dev = self.owner.device
        out = torch.empty((len(self.keep[1]), nfull * bw, 3), dtype=torch.float32, device=dev)       # the caller's stream owns the result
        main = torch.cuda.current_stream(dev)
        if self.lanes == 1:
            for j in range(nfull):
                self.specs[0].copy_(specs[j * bw:(j + 1) * bw])
                self.graphs[0].replay()
                out[:, j * bw:(j + 1) * bw].copy_(self.probs[0])
            return out, (tail() if tail is not None else None)
        for i in range(min(self.lanes, nfull)):
            self.side[i].wait_stream(main)
        for j in range(nfull):                                        # issued round-robin so that every lane has work queued early
            i = j % self.lanes
            with torch.cuda.stream(self.side[i]):
                self.specs[i].copy_(specs[j * bw:(j + 1) * bw])
                self.graphs[i].replay()
                out[:, j * bw:(j + 1) * bw].copy_(self.probs[i])
        t = tail() if tail is not None else None
        for i in range(min(self.lanes, nfull)):
            main.wait_stream(self.side[i])
        return out, t


SERVING_LANES = 4      # batches in flight side by side in serve_spectrograms / streaming_inferences (at most: see lane_budget)


def lane_budget(batch, device=None):
    """How many lanes of `batch` clips may run side by side.  The paired kernels of the 2x2-image blocks hold their CU while they wait for their
    partner workgroup (include/mkws.h, failure contract): all lanes' pairs TOGETHER must fit the chip -- 2 workgroups per 8 clips -- or halves
    of different lanes can fill the CUs their partners are waiting for (six lanes of 256 clips: 384 such workgroups for 256 CUs measured 23 ms
    per 2950 windows instead of 3.7, every wait running into the kernels' timeout; profiles/r06_notes.md section 8)."""
    import torch
    cus = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).multi_processor_count
    return max(1, cus // (2 * ((int(batch) + 7) // 8)))


def serve_spectrograms(emb_model, heads, specs, batch_windows=4096, use_graph=True, graphs_used=None):
    """CUDA spectrograms [windows, 49, 40] -> CUDA softmax outputs [n_heads, windows, 3]: the embedding in batches of
    min(batch_windows, max_batch) windows, every head on every batch.  Asynchronous on the caller's current stream (no host copy, no
    synchronisation): what streaming_inferences runs per chunk and what `bench.py --config stream` times."""
    import torch
    from ..head import Head
    bw = min(batch_windows, emb_model.max_batch)
    nwin = specs.shape[0]
    nfull = nwin // bw if (use_graph and tuple(specs.shape[1:]) == (49, 40)) else 0

    def eager_from(s0):                                       # launch by launch on the caller's own handle: [N, windows, 3] per batch
        return [Head.forward_many(heads, emb_model.forward(specs[s:s + bw])) for s in range(s0, nwin, bw)]
    if nfull == 0:
        parts = eager_from(0)
        return torch.cat(parts, dim=1) if parts else torch.zeros((len(heads), 0, 3), dtype=torch.float32, device=emb_model.device)
    # full batches replay captured graphs, up to SERVING_LANES of them side by side (see _BatchGraph); one lane = the same launches on
    # the same plan as the eager path, bit for bit; several lanes = the workgroup shapes of the clips they hold together
    # (the cluster kernel of small handles spins on up to 14 co-resident members per launch: concurrent lanes could starve
    # each other of CUs, include/mkws.h -- such handles replay one batch at a time)
    lanes = 1 if emb_model.get_option("fuse_cluster") else min(SERVING_LANES, nfull, lane_budget(bw, emb_model.device))
    if lanes > 1:
        lanes = len(concurrent_streams(lanes, emb_model.device))     # (fewer when the runtime has fewer hardware queues to give)
    bg = _BatchGraph.get(emb_model, heads, bw, lanes)
    if graphs_used is not None:
        graphs_used.setdefault(bg, bg.heals)
    full, rest = bg.run_all(specs, nfull, (lambda: eager_from(nfull * bw)) if nfull * bw < nwin else None)
    return torch.cat([full] + list(rest), dim=1) if rest else full


def streaming_inferences(models, model_settings, audio, sample_rate=16000, clip_duration_ms=1000, clip_stride_ms=20,
                         batch_windows=4096, max_chunk_length_sec=None, use_graph=True, _retry=True):
    """Softmax outputs for every window.  `models`: one TransferLearnedModel or a list sharing one embedding
    (multi-keyword serving: the EfficientNet forward runs once, each keyword adds only its 18.5 k-parameter
    head).  Returns [num_windows, 3] (or a list of them)."""
    import torch
    single = not isinstance(models, (list, tuple))
    mlist = [models] if single else list(models)
    clip = int(clip_duration_ms * sample_rate / 1000)
    stride = int(clip_stride_ms * sample_rate / 1000)
    outs = [[] for _ in mlist]
    emb_model = mlist[0].embedding
    graphs_used = {}                           # graph -> its heal count when this stream first used it

    def degraded():                            # times the handles of this stream have left the exchange kernels after a failure
        return sum(e.get_option("pair_degraded") for e in [emb_model] + list(getattr(emb_model, "_replicas", [])))
    degraded0 = degraded()
    from ..head import Head
    audio_arr = audio if torch.is_tensor(audio) else np.asarray(audio, dtype=np.float32)
    max_chunk = None if max_chunk_length_sec is None else int(max_chunk_length_sec * sample_rate)
    for chunk in chunk_audio(audio_arr, max_chunk):
        specs = stream_spectrograms(model_settings, chunk, clip, stride)
        heads = [m.head for m in mlist]
        probs = serve_spectrograms(emb_model, heads, specs, batch_windows, use_graph, graphs_used)
        for k in range(len(mlist)):
            outs[k].append(probs[k])
    if graphs_used:
        # a failed exchange inside a replay leaves NaN rows and no return code: look at the handles once everything has run, and redo
        # the stream on the healed handles (the first run() of the repeat re-captures; a healed handle cannot fail again).  A heal in
        # the middle of the stream -- by a re-capture or by the eager call of a ragged tail -- means earlier batches were poisoned.
        torch.cuda.synchronize(emb_model.device)
        if degraded() != degraded0 or any(g.exchange_failed() or g.heals != h0 for g, h0 in graphs_used.items()):
            if not _retry:
                # the repeat ran with every handle of the stream off the exchange kernels (below): nothing left that could fail this way
                from .._lib import MKWS_ERR_EXCHANGE, MkwsError
                raise MkwsError(MKWS_ERR_EXCHANGE, "an in-kernel exchange failed again while the stream was being repeated on the single-workgroup kernels")
            import warnings
            warnings.warn("multilingual_kws_amd: an in-kernel exchange failed during a graph replay; repeating the stream on the single-workgroup kernels. "
                          "This embedding handle and its serving replicas STAY on that plan for the rest of the process (the library retires a handle's "
                          "exchange kernels for good once one exchange has failed: include/mkws.h, MKWS_ERR_EXCHANGE); create a new handle to get them back",
                          RuntimeWarning)
            # EVERY serving-lane replica leaves the exchange kernels before the repeat, not only the handle whose error word was set: a
            # replica that fails during the repeat would return its NaN rows with nobody looking
            for e in [emb_model] + list(getattr(emb_model, "_replicas", [])):
                if e.get_option("exchange_error"):                              # heals: the wrapper repeats the call that reports the error.  A known
                    e.forward(torch.zeros((1, 49, 40), dtype=torch.float32, device=emb_model.device))   # one-window batch, not the last chunk's (maybe empty) one
                e.set_option("fuse_pair", 0)
                e.set_option("fuse_cluster", 0)
            _BatchGraph.forget(emb_model)                                       # graphs captured on the old plan
            return streaming_inferences(models, model_settings, audio, sample_rate, clip_duration_ms, clip_stride_ms, batch_windows, max_chunk_length_sec,
                                        use_graph, _retry=False)
    res = [torch.cat(o).cpu().numpy() if o else np.zeros((0, 3), np.float32) for o in outs]
    return res[0] if single else res


class StreamingSession:
    """Live serving of the reference's window loop (batch_streaming_analysis.py:99-117), one or a few windows at a time: the
    newest `batch` one-second windows -> micro-frontend -> embedding -> every keyword head, as ONE hipGraph replay.

    At batch 1 the path is ~65 small launches; issued one by one each costs a host round trip, so a window took 0.55 ms with
    the GPU mostly idle.  All mkws_* calls are asynchronous, allocation-free and synchronisation-free on the caller's stream
    (include/mkws.h), so the whole chain is captured once and replayed: the per-window host cost is one copy into the static
    input and one graph launch.  Results are the same launches on the same buffers, i.e. bit-identical to the eager calls.

    models: TransferLearnedModel(s) sharing one embedding (or pass embedding= and heads= directly)."""

    def __init__(self, models=None, model_settings=None, batch=1, embedding=None, heads=None, use_graph=True):
        import torch
        from ..head import Head
        if models is not None:
            mlist = list(models) if isinstance(models, (list, tuple)) else [models]
            embedding, heads = mlist[0].embedding, [m.head for m in mlist]
        self.embedding, self.heads, self.batch = embedding, list(heads), int(batch)
        if self.batch > embedding.max_batch:
            raise ValueError(f"StreamingSession(batch={batch}) exceeds the embedding handle's max_batch={embedding.max_batch}")
        ms = model_settings or input_data.standard_microspeech_model_settings(3)
        self.samples = ms["desired_samples"]
        self.fe = input_data._frontend_for(ms, self.samples)
        dev = embedding.device
        self.audio = torch.zeros((self.batch, self.samples), dtype=torch.float32, device=dev)      # static graph input
        self._Head = Head
        self.graph = None
        self.recaptures = 0
        self.probs = self._chain()                      # eager pass: creates every lazily-built table / attribute
        if use_graph:
            self._capture()

    def _capture(self):
        import torch
        dev = self.embedding.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._chain()                               # (heals a handle with a recorded exchange failure: the wrapper repeats the call)
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.probs = self._chain()
        self.graph = g

    def _chain(self):
        if self.batch == 1:
            # one window: the frame-parallel streaming kernels (49 frames on 49 waves across the chip, then the window's scan) instead of
            # the batch kernel's one workgroup walking the 49 frames four at a time: 36 -> ~10 us; bit-identical (tests/test_frontend_gpu.py)
            spec = self.fe.stream(self.audio[0], self.samples, self.samples)
        else:
            spec = self.fe.forward(self.audio)
        return self._Head.forward_many(self.heads, self.embedding.forward(spec))

    def infer(self, audio):
        """audio: [samples] or [batch, samples] float32 (numpy, CPU or CUDA tensor) -> CUDA tensor [n_heads, batch, 3] of softmax
        outputs (a view of the session's static output: valid until the next infer)."""
        import torch
        a = audio if torch.is_tensor(audio) else torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
        self.audio.copy_(a.reshape(self.batch, self.samples), non_blocking=True)
        if self.graph is not None:
            if self.embedding.get_option("exchange_error"):
                # the PREVIOUS replay ran a failed pair / cluster exchange (its output was all-NaN, never plausible numbers); the captured
                # launches would poison every later window as well: heal the handle and capture the single-workgroup kernels
                self._capture()
                self.recaptures += 1
            self.graph.replay()
        else:
            self.probs = self._chain()
        return self.probs


def detect(inferences, flags: StreamFlags, threshold, sample_rate=16000, data_samples=None):
    """Runs the detector over per-window softmax outputs; returns (found_words, found_words_w_confidences)
    exactly as the reference collects them (:143-167)."""
    clip = int(flags.clip_duration_ms * sample_rate / 1000)
    stride = int(flags.clip_stride_ms * sample_rate / 1000)
    offsets = window_offsets(data_samples, clip, stride) if data_samples is not None else [i * stride for i in range(len(inferences))]
    element = RecognizeResult()
    rc = SingleTargetRecognizeCommands(labels=flags.labels(), average_window_duration_ms=flags.average_window_duration_ms,
                                       detection_threshold=threshold, suppression_ms=flags.suppression_ms,
                                       minimum_count=flags.minimum_count, target_id=2)
    found, found_conf = [], []
    for ix, off in enumerate(offsets):
        t_ms = int(off * 1000 / sample_rate)
        rc.process_latest_result(inferences[ix], t_ms, element)
        if element.is_new_command and element.found_command != "_silence_":
            found.append([element.found_command, t_ms])
            found_conf.append([element.found_command, t_ms, element.score])
    return found, found_conf


def calculate_streaming_accuracy(model, model_settings, flag_list, existing_inferences=None):
    """Reference signature (:50-179): one wav, several StreamFlags; returns (results, inferences) with
    results = [(FLAGS, {threshold: (found_words, found_words_w_confidences)})]."""
    assert len(set([f.wav for f in flag_list])) == 1, "can only process one wav"
    assert len(set([f.clip_duration_ms for f in flag_list])) == 1, "cannot vary"
    assert len(set([f.clip_stride_ms for f in flag_list])) == 1, "cannot vary"
    with open(flag_list[0].wav, "rb") as f:
        audio, sample_rate = input_data.decode_wav(f.read())
    if existing_inferences is not None:
        inferences = existing_inferences
    else:
        inferences = streaming_inferences(model, model_settings, audio, sample_rate, flag_list[0].clip_duration_ms,
                                          flag_list[0].clip_stride_ms, max_chunk_length_sec=flag_list[0].max_chunk_length_sec)
    results = []
    for FLAGS in flag_list:
        res_thresh = {}
        for threshold in FLAGS.detection_thresholds:
            res_thresh[threshold] = detect(inferences, FLAGS, threshold, sample_rate, data_samples=audio.shape[0])
        results.append((FLAGS, res_thresh))
    return results, inferences


@dataclass
class StreamTarget:
    """Reference :188-195 -- one keyword's streaming evaluation: where its model is, what to run it on, where results go."""
    target_lang: str
    target_word: str
    model_path: os.PathLike
    stream_flags: List[StreamFlags]
    destination_result_pkl: Optional[os.PathLike] = None
    destination_result_inferences: Optional[os.PathLike] = None


def eval_stream_test(st: StreamTarget, live_model=None):
    """Reference :198-241.  -> {target_word: [(FLAGS, {threshold: (found_words, found_words_w_confidences)}), ...]}, or None (after a
    message) when destination_result_pkl already exists.  The model is `live_model` or TransferLearnedModel.load(st.model_path) (the
    directory transfer_learn's model.save() wrote -- the counterpart of tf.keras.models.load_model).  Results are pickled to
    destination_result_pkl and the raw per-window softmax outputs saved (np.save) to destination_result_inferences when those are
    given; inferences found there are re-used instead of being recomputed.  (The reference reads them back from the PICKLE path --
    np.load(st.destination_result_pkl), a file it has just established does not exist -- so its re-use branch cannot run; here the
    branch loads the file it tested for.)"""
    if live_model is not None:
        model = live_model
    else:
        from .transfer_learning import TransferLearnedModel
        model = TransferLearnedModel.load(os.fspath(st.model_path))
    model_settings = input_data.standard_microspeech_model_settings(label_count=3)

    if st.destination_result_pkl is not None and os.path.isfile(st.destination_result_pkl):
        print("results already present", st.destination_result_pkl, flush=True)
        return
    loaded_inferences = None
    if st.destination_result_inferences is not None and os.path.isfile(st.destination_result_inferences):
        print("inferences already present", flush=True)
        loaded_inferences = np.load(st.destination_result_inferences)

    results = {}
    results[st.target_word], inferences = calculate_streaming_accuracy(model, model_settings, st.stream_flags, loaded_inferences)

    if st.destination_result_pkl is not None:
        print("SAVING results TO\n", st.destination_result_pkl)
        with open(st.destination_result_pkl, "wb") as fh:
            pickle.dump(results, fh)
    if loaded_inferences is None and st.destination_result_inferences is not None:
        print("SAVING inferences TO\n", st.destination_result_inferences, flush=True)
        np.save(st.destination_result_inferences, inferences)
    return results


def multi_keyword_detections(keywords, models, wav, detection_threshold=0.9, inference_chunk_len_seconds=1200, groundtruth=None,
                             average_window_duration_ms=100, suppression_ms=500, write_detections=None):
    """The detections dict of run.py:89-152 for N keywords from ONE pass over the recording.

    The reference loads one full Keras model per keyword and repeats the window loop, the micro-frontend and the EfficientNet forward
    for each (one child process per keyword); here `models` (TransferLearnedModels sharing one embedding, transfer_learning.
    load_models_shared) are N 18.5 k-parameter heads on one embedding pass (streaming_inferences), and each keyword's detector runs over
    its own head's outputs.  Per keyword the detections are exactly what eval_stream_test yields for StreamFlags(detection_thresholds=
    [detection_threshold], average_window_duration_ms=100, suppression_ms=500, max_chunk_length_sec=inference_chunk_len_seconds);
    they are merged and sorted by time (stable, like the reference's sorted()).  -> dict(keywords=..., detections=[dict(keyword, time_ms,
    confidence, groundtruth)], min_threshold=...): groundtruth "ng" without a ground-truth file, otherwise tpr_fpr.get_groundtruth's
    classification against its rows `keyword,time_ms` (as shipped: first keyword only).  Also written as JSON to write_detections."""
    import json
    keywords, models = list(keywords), list(models)
    if len(models) != len(keywords) or len(set(keywords)) != len(keywords):
        raise ValueError(f"discrepancy: {len(models)} models provided for {len(set(keywords))} keywords")
    if inference_chunk_len_seconds <= 0:
        raise ValueError("inference_chunk_len_seconds must be positive")
    with open(wav, "rb") as f:
        audio, sample_rate = input_data.decode_wav(f.read())
    model_settings = input_data.standard_microspeech_model_settings(label_count=3)
    per_keyword = [None] * len(models)
    by_embedding = {}
    for i, m in enumerate(models):                                  # one pass per distinct embedding (normally one)
        by_embedding.setdefault(id(m.embedding), []).append(i)
    for idxs in by_embedding.values():
        got = streaming_inferences([models[i] for i in idxs], model_settings, audio, sample_rate, 1000, 20,
                                   max_chunk_length_sec=inference_chunk_len_seconds)
        for i, inf in zip(idxs, got):
            per_keyword[i] = inf
    unsorted_detections = []
    for keyword, inferences in zip(keywords, per_keyword):
        flags = StreamFlags(wav=wav, ground_truth=groundtruth, target_keyword=keyword, detection_thresholds=[detection_threshold],
                            average_window_duration_ms=average_window_duration_ms, suppression_ms=suppression_ms, time_tolerance_ms=750,
                            max_chunk_length_sec=inference_chunk_len_seconds)
        unsorted_detections.extend(detect(inferences, flags, detection_threshold, sample_rate, data_samples=audio.shape[0])[1])
    detections_with_confidence = sorted(unsorted_detections, key=lambda d: d[1])
    if groundtruth is None:
        detections_with_confidence = [dict(keyword=d[0], time_ms=d[1], confidence=d[2], groundtruth="ng") for d in detections_with_confidence]
    else:
        import csv
        from .tpr_fpr import get_groundtruth
        with open(groundtruth, "r") as fh:
            groundtruth_data = [(row[0], float(row[1])) for row in csv.reader(fh) if row]
        detections_with_confidence = get_groundtruth(detections_with_confidence, keywords, groundtruth_data)
    detections = dict(keywords=keywords, detections=detections_with_confidence, min_threshold=detection_threshold)
    if write_detections is not None:
        with open(write_detections, "w") as fh:
            json.dump(detections, fh)
    return detections
