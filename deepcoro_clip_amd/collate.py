"""Batch layout of the contrastive runner's input side (reference: dataloaders/video_clip_dataset.py:843-969
`custom_collate_fn`, :638-644 pad-to-512 tokenisation).

Same batch dict as the reference's collate — `videos`, `encoded_texts{input_ids, attention_mask}`, `paths`, `reports` and,
in multi-positive mode, `positive_mask`, `positive_weights`, `text_ids`, `text_metadata` — so `runner._preprocess_inputs`
and `_run_epoch` consume it unchanged.  Two input-side changes that exist because of the MI355X path:

  * frames may stay **uint8** (`[.., T, H, W, 3]` raw pixels): they are uploaded as bytes (4x less PCIe traffic than the
    reference's float32 host pipeline, utils/video.py:278-288) and normalised on the GPU by `dcc_u8_normalize`
    (runner._preprocess_inputs with config.video_mean / video_std);
  * **text-length bucketing** (`trim_text_padding`): the reference pads every report to 512 tokens, which makes BERT cost
    as much as the video tower (96.6 vs 128.5 GFLOP); here a batch is cut to its longest valid sequence,
    rounded up to `bucket` tokens.  Padded keys are masked in attention, so the CLS embedding is unchanged.
"""
from __future__ import annotations

from typing import Any, Dict, List, Sequence

import numpy as np
import torch


def _as_video_tensor(v) -> torch.Tensor:
    t = torch.from_numpy(v) if isinstance(v, np.ndarray) else torch.as_tensor(v)
    return t if t.dtype == torch.uint8 else t.float()


def _stack_videos(videos: Sequence) -> torch.Tensor:
    return torch.stack([_as_video_tensor(v) for v in videos])  # [B, T, H, W, C] or [B, N, T, H, W, C]


def trim_padding(input_ids: torch.Tensor, attention_mask: torch.Tensor, bucket: int = 16):
    """Cut right padding: keep max(valid length) rounded up to a multiple of `bucket` (never beyond the given width)."""
    if input_ids.numel() == 0:
        return input_ids, attention_mask
    longest = int(attention_mask.to(torch.int64).sum(dim=1).max().item())
    keep = min(input_ids.shape[1], max(bucket, -(-longest // bucket) * bucket))
    return input_ids[:, :keep].contiguous(), attention_mask[:, :keep].contiguous()


def custom_collate_fn(batch, trim_text_padding: bool = False, bucket: int = 16) -> Dict[str, Any]:
    """batch: sequence of (video array, payload, path, raw_text) samples as the reference dataset yields them."""
    videos, payloads, paths, raw_texts = zip(*batch)
    multi = isinstance(payloads[0], dict) and "positive_entries" in payloads[0]
    videos_tensor = _stack_videos(videos)
    if not multi:
        texts = None
        if isinstance(payloads[0], dict) and "input_ids" in payloads[0]:
            ids = torch.stack([p["input_ids"] for p in payloads])
            mask = torch.stack([p["attention_mask"] for p in payloads])
            if trim_text_padding:
                ids, mask = trim_padding(ids, mask, bucket)
            texts = {"input_ids": ids, "attention_mask": mask}
        return {"videos": videos_tensor, "encoded_texts": texts, "paths": list(paths), "reports": list(raw_texts)}

    # multi-positive mode: the batch's UNIQUE prompts become the text side ([M, S]); mask / weights say which video owns which
    column: Dict[str, int] = {}
    ids_rows: List[torch.Tensor] = []
    mask_rows: List[torch.Tensor] = []
    meta: List[Dict[str, Any]] = []
    for payload in payloads:
        for entry in payload.get("positive_entries", []):
            tid = str(entry.get("text_id"))
            if tid in column:
                continue
            column[tid] = len(column)
            enc = entry["encoding"]
            ids_rows.append(enc["input_ids"].clone())
            mask_rows.append(enc["attention_mask"].clone())
            meta.append({"text_id": tid, "prompt_text": entry.get("prompt_text", ""), "prompt_type": entry.get("prompt_type"),
                         "soft_weight": float(entry.get("soft_weight", 1.0)), "edge_weight": float(entry.get("edge_weight", 1.0)),
                         "is_abnormal": bool(entry.get("is_abnormal", False)), "tags": entry.get("tags")})
    if not column:  # no prompt anywhere: one blank column built from the first sample's first encoding (reference :885-900)
        first = payloads[0]["positive_entries"][0]["encoding"]
        column["__blank__"] = 0
        ids_rows.append(first["input_ids"].clone())
        mask_rows.append(first["attention_mask"].clone())
        meta.append({"text_id": "__blank__", "prompt_text": "", "prompt_type": None, "soft_weight": 1.0, "edge_weight": 1.0,
                     "is_abnormal": False, "tags": None})
    input_ids, attention_mask = torch.stack(ids_rows), torch.stack(mask_rows)
    if trim_text_padding:
        input_ids, attention_mask = trim_padding(input_ids, attention_mask, bucket)
    B, M = len(videos), input_ids.shape[0]
    positive_mask = torch.zeros(B, M, dtype=torch.float32)
    positive_weights = torch.zeros(B, M, dtype=torch.float32)
    for row, payload in enumerate(payloads):
        for entry in payload.get("positive_entries", []):
            col = column.get(str(entry.get("text_id")))
            if col is None:
                continue
            positive_mask[row, col] = 1.0
            positive_weights[row, col] = float(entry.get("edge_weight", 1.0)) * float(entry.get("soft_weight", 1.0))
    reports = []
    for payload, fallback in zip(payloads, raw_texts):
        pos = payload.get("positive_entries", [])
        reports.append(pos[0].get("prompt_text", fallback) if pos else fallback)
    return {"videos": videos_tensor, "encoded_texts": {"input_ids": input_ids, "attention_mask": attention_mask},
            "positive_mask": positive_mask, "positive_weights": positive_weights,
            "text_ids": [t if t != "__blank__" else None for t in column], "text_metadata": meta, "paths": list(paths),
            "reports": reports}
