"""-m gpu: a tripwire under the fused kernels (round-4 verdict, "an unchased device fault").

Every forward pass here runs with (a) the caller's input between two NaN regions -- an under- or over-read that reaches a result changes it --,
(b) the caller's output between two canary regions -- a store outside [B, 1024] is seen --, and (c) the handle in guard-band mode
(MKWS_EMBED_GUARD, include/mkws.h): its workspace starts as a NaN canary pattern and every sub-buffer carved from it is fenced by bands that
must still hold the canary afterwards.  The results must equal, bit for bit, those of an ordinary handle on ordinary buffers: nothing reads
workspace it has not written, nothing reads or writes outside its rows.  Batch sizes: full and ragged workgroups of both chain
instantiations (1024 / 1023: 4-clip workgroups and 8-clip pairs; 512 / 511: 2-clip workgroups and 4-clip pairs) and the cluster-kernel
handles (1, 3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CANARY = 0x7FC00BAD
PAD = 1 << 15          # floats in front of and behind every caller-owned buffer


def _fenced(n):
    """A CUDA buffer [PAD | n | PAD] -> (whole int32 view, float32 view of the middle); everything starts as the NaN canary."""
    whole = torch.full((PAD + n + PAD,), CANARY, dtype=torch.int32, device="cuda")
    return whole, whole[PAD:PAD + n].view(torch.float32)


def _pads_intact(whole, n):
    return bool((whole[:PAD] == CANARY).all()) and bool((whole[PAD + n:] == CANARY).all())


@pytest.fixture(scope="module")
def blob():
    from multilingual_kws_amd import weights
    return weights.synthetic_blob()


def _specs(B):
    from multilingual_kws_amd import synth
    from multilingual_kws_amd.frontend import Frontend
    n = min(B, 64)
    s = Frontend().forward(torch.from_numpy(synth.clips_float32(n)).cuda())
    return s.repeat((B + n - 1) // n, 1, 1)[:B].contiguous() if B > n else s


@pytest.mark.parametrize("B,opts", [(1, {}), (3, {}), (3, {"fuse_cluster": 0}), (511, {}), (512, {}), (512, {"fuse_chain": 0}), (1023, {}),
                                     (1024, {}), (1024, {"fuse_chain": 0}), (1024, {"fuse_pair": 0}), (1024, {"fuse_top": 0, "fuse_mid": 2})])
def test_embedding_forward_stays_inside_its_buffers(blob, B, opts, monkeypatch):
    from multilingual_kws_amd.embedding_model import EmbeddingModel
    spec = _specs(B)
    plain = EmbeddingModel(blob, max_batch=B)
    monkeypatch.setenv("MKWS_EMBED_GUARD", "4096")
    guarded = EmbeddingModel(blob, max_batch=B)
    monkeypatch.delenv("MKWS_EMBED_GUARD")
    assert plain.get_option("guard_floats") == 0 and plain.get_option("guard_violations") == 0
    assert guarded.get_option("guard_floats") == 4096 and guarded.get_option("guard_bands") >= 15
    for em in (plain, guarded):
        for k, v in opts.items():
            em.set_option(k, v)
    want = plain.forward(spec)
    win, sview = _fenced(B * 1960)
    wout, eview = _fenced(B * 1024)
    sview.copy_(spec.reshape(-1))
    for rep in range(3):                     # repeated calls: flags / generation counters / rings of the exchange kernels wrap around
        got = guarded.forward(sview.view(B, 49, 40), out=eview.view(B, 1024))
        torch.cuda.synchronize()
        assert got.data_ptr() == eview.data_ptr()
        assert _pads_intact(wout, B * 1024), "a store outside the caller's [B, 1024] output"
        assert _pads_intact(win, B * 1960), "a store into the caller's input buffer's surroundings"
        assert guarded.get_option("guard_violations") == 0, "a store into a guard band of the handle's workspace"
        assert torch.isfinite(got).all()
        assert torch.equal(got, want), "results depend on memory outside the buffers (NaN-filled workspace / NaN-padded input)"
    if guarded.max_batch > 1:                # a smaller batch on the same handle: rows B' .. max_batch of every buffer stay unread
        Bs = max(1, B // 2 - 1)
        eview.view(torch.int32).fill_(CANARY)
        got = guarded.forward(sview[:Bs * 1960].view(Bs, 49, 40), out=eview[:Bs * 1024].view(Bs, 1024))
        torch.cuda.synchronize()
        assert torch.equal(got, want[:Bs]) and bool((eview.view(torch.int32)[Bs * 1024:] == CANARY).all())
        assert guarded.get_option("guard_violations") == 0


@pytest.mark.parametrize("B", [1, 5, 1023, 1024])
def test_frontend_stays_inside_its_buffers(B):
    from multilingual_kws_amd import synth
    from multilingual_kws_amd.frontend import Frontend
    fe = Frontend()
    n = min(B, 96)
    a = torch.from_numpy(synth.clips_float32(n)).cuda()
    audio = a.repeat((B + n - 1) // n, 1)[:B].contiguous()
    want = fe.forward(audio)
    win, aview = _fenced(B * 16000)
    wout, sview = _fenced(B * 1960)
    aview.copy_(audio.reshape(-1))
    got = fe.forward(aview.view(B, 16000), out=sview.view(B, 49, 40))
    torch.cuda.synchronize()
    assert _pads_intact(wout, B * 1960) and _pads_intact(win, B * 16000)
    assert torch.equal(got, want)
    pcm = (audio * 32768).to(torch.int16)                      # int16 input
    assert torch.equal(fe.forward(pcm), want)
    if B == 1:                                                  # the one-window streaming kernels of the live path
        assert torch.equal(fe.stream(aview, 16000, 16000), want)
        assert _pads_intact(win, B * 16000)
