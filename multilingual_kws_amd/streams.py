"""HIP streams that really run side by side.

Independent batches (the serving lanes of embedding.batch_streaming_analysis) only overlap if their streams sit on DIFFERENT hardware
queues.  The HIP runtime deals streams onto GPU_MAX_HW_QUEUES queues (default 4) in an order that depends on every stream the process has
created so far -- PyTorch alone keeps two pools of 32; with four queues fresh streams k and 7 - k share one -- and two streams on one queue
run their launches one after the other: four 256-clip lanes measured 1.49 ms where four distinct queues give 0.99 ms
(profiles/r06_notes.md section 8).  There is no API that tells a stream's queue, so this module MEASURES: a candidate stream is accepted
when a short spin kernel on it overlaps the same kernel on every stream accepted before it.  (More queues are not better: with
GPU_MAX_HW_QUEUES = 8 / 16 the same four lanes serve 800 k windows/s against 885 k on the default four, same call.)"""
import time

_cache = {}
_exhausted = set()              # devices whose search has run out of candidates: the answer stays what it is (a search costs tens of ms)
_SPIN_CYCLES = 400_000          # ~0.2 ms at the shader clock: long against launch latency, short against anything a caller would notice
_CANDIDATES = 12


def _spin_all(torch, streams, dev):
    """Wall time of one spin kernel on each of `streams`, all released together."""
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(_SPIN_CYCLES)
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


def concurrent_streams(n, device=None):
    """Up to n torch.cuda.Stream objects on `device` whose launches overlap pairwise (fewer if the runtime has fewer queues: callers take
    len() of the result as their lane count).  Measured once per device; the streams are kept for the life of the process."""
    import torch
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    have = _cache.setdefault(key, [])
    if len(have) >= n or key in _exhausted:
        return have[:n]
    if not hasattr(torch.cuda, "_sleep"):                                         # no spin kernel in this torch build: fresh streams, unmeasured
        while len(have) < n:
            have.append(torch.cuda.Stream(device=dev))
        return have[:n]
    with torch.cuda.device(dev):
        first = have[0] if have else torch.cuda.Stream(device=dev)
        _spin_all(torch, [first], dev)                                        # (first launch of the spin kernel: module load)
        one = min(_spin_all(torch, [first], dev) for _ in range(3))
        if not have:
            have.append(first)
        tried = 0
        while len(have) < n and tried < _CANDIDATES:
            cand = torch.cuda.Stream(device=dev)
            tried += 1
            # against every accepted stream, one pair at a time: side by side = about one spin (1.0-1.3 measured), a shared queue = two.
            # Two trials per pair, the better one counts (a stray host hiccup must not reject a stream)
            if all(min(_spin_all(torch, [s, cand], dev) for _ in range(2)) < 1.6 * one for s in have):
                have.append(cand)
        if len(have) < n:
            _exhausted.add(key)
    return have[:n]

