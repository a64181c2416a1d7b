"""Digest of a step's detections: what bench.py prints as `results_digest` and what the two-stream parity test compares.

A step returns, per frame, {'pred_boxes' [n, 7] f32, 'pred_scores' [n] f32, 'pred_labels' [n] i64} (the reference's
post_processing record, detector3d_template.py:222-343). The digest is the per-frame box count plus a BLAKE2 hash over the raw
bytes of the three arrays, frame after frame: two runs agree on it exactly when every kept box, score and label is bit-identical.
"""
import hashlib

import numpy as np


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def frame_digest(res):
    """hex digest of one frame's record"""
    h = hashlib.blake2b(digest_size=8)
    for key, dt in (("pred_boxes", np.float32), ("pred_scores", np.float32), ("pred_labels", np.int64)):
        h.update(np.ascontiguousarray(_np(res[key]).astype(dt, copy=False)).tobytes())
    return h.hexdigest()


def step_digest(results):
    """(per-frame box counts, per-frame hex digests, one hex digest of the whole step)"""
    counts = [int(_np(r["pred_boxes"]).shape[0]) for r in results]
    frames = [frame_digest(r) for r in results]
    h = hashlib.blake2b(digest_size=8)
    for c, f in zip(counts, frames):
        h.update(("%d:%s;" % (c, f)).encode())
    return counts, frames, h.hexdigest()
