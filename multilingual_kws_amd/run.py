"""Drop-in for the `inference` entry point of multilingual_kws/run.py (:24-152): keyword detections over a long recording.

Same arguments and the same detections.json; what differs is underneath: the reference starts one child process per keyword, each
loading a full Keras model and repeating the window loop, the micro-frontend and the EfficientNet forward; here the saved few-shot
models are loaded as N heads on ONE shared embedding (transfer_learning.load_models_shared) and the recording is walked once
(embedding.batch_streaming_analysis.multi_keyword_detections).  The browser visualizer (run.py:154-260) and the `train` command's
file bookkeeping are outside the hot path: visualizer=True raises NotImplementedError; fine-tuning is transfer_learning.transfer_learn."""
import os
from pathlib import Path
from typing import List, Optional

from .embedding import batch_streaming_analysis as sa
from .embedding import transfer_learning


def eval(streamtarget: sa.StreamTarget, results: dict):
    """run.py:20-21."""
    results.update(sa.eval_stream_test(streamtarget))


def inference(
    keywords: List[str],
    modelpaths,
    wav: os.PathLike,
    groundtruth: Optional[os.PathLike] = None,
    transcript: Optional[os.PathLike] = None,
    visualizer: bool = False,
    serve_port: int = 8080,
    detection_threshold: float = 0.9,
    inference_chunk_len_seconds: int = 1200,
    language: str = "unspecified_language",
    write_detections: Optional[os.PathLike] = None,
    overwrite: bool = False,
):
    """Runs inference on a streaming audio file; arguments as the reference's (keywords: list of target words, or one word; modelpaths:
    comma-delimited string of saved few-shot model directories -- a list of paths or of live TransferLearnedModels is accepted too;
    detection_threshold: detector threshold; inference_chunk_len_seconds: chunk length of the recording; write_detections: where to
    write detections.json).  Returns the detections dict (the reference returns None and only writes the file)."""
    if isinstance(keywords, str) or len(keywords[0]) == 1:
        print(f"NOTE - assuming a single keyword was passed in: {keywords}")
        keywords = ["".join(keywords)]
    print(f"Target keywords: {keywords}")
    if isinstance(modelpaths, str):
        modelpaths = modelpaths.split(",")
    modelpaths = list(modelpaths)
    assert len(modelpaths) == len(set(keywords)), f"discrepancy: {len(modelpaths)} modelpaths provided for {len(set(keywords))} keywords"
    live = [m for m in modelpaths if hasattr(m, "predict")]
    for p in modelpaths:
        assert hasattr(p, "predict") or os.path.exists(p), f"{p} inference model not found"
    assert os.path.exists(wav), f"{wav} streaming audio wavfile not found"
    assert Path(wav).suffix == ".wav", f"{wav} filetype not supported"
    assert inference_chunk_len_seconds > 0, "inference_chunk_len_seconds must be positive"
    if visualizer:
        raise NotImplementedError("the browser visualizer of run.py is outside this build's scope; detections are returned / written as JSON")
    print(f"performing inference using detection threshold {detection_threshold}")
    models = live if len(live) == len(modelpaths) else transfer_learning.load_models_shared(modelpaths)
    detections = sa.multi_keyword_detections(keywords, models, wav, detection_threshold=detection_threshold,
                                             inference_chunk_len_seconds=inference_chunk_len_seconds, groundtruth=groundtruth,
                                             write_detections=write_detections)
    for d in detections["detections"]:
        print(d)
    return detections
