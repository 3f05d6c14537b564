"""Input side: raw uint8 frames normalised inside the patch-embedding gather, text-length bucketing."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_uint8_patch_gather_equals_host_normalisation_then_fp32_path():
    """dcc_im2col3d_u8 on bytes == dcc_im2col3d on (x - mean) / std computed on the host (utils/video.py:278-288), bit for
    bit in fp32 and bf16 rows; padding taps are zeros of the normalised tensor."""
    from deepcoro_clip_amd import kernels as K

    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (2, 4, 32, 32, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor([110.2, 101.5, 95.25])
    std = torch.tensor([60.0, 58.5, 62.25])
    host = ((frames.float() - mean) / std).contiguous()
    for dt in (torch.float32, torch.bfloat16):
        ref = K.im2col3d(host.cuda(), (3, 7, 7), (2, 4, 4), (1, 3, 3), 448, dt)
        got = K.im2col3d(frames.cuda(), (3, 7, 7), (2, 4, 4), (1, 3, 3), 448, dt, norm=(mean.cuda(), std.cuda()))
        # the kernel multiplies by 1/std where the host divides: at most one ulp apart in fp32
        assert (got.float() - ref.float()).abs().max() <= (2e-7 * 5 if dt == torch.float32 else 4e-2)
        assert torch.equal(got[:, 441:], torch.zeros_like(got[:, 441:]))  # padding of the column index
    with pytest.raises(Exception):
        K.im2col3d(frames.cuda(), (3, 7, 7), (2, 4, 4), (1, 3, 3), 448, torch.float32)


def test_video_encoder_takes_uint8_frames_and_runner_keeps_bytes():
    from deepcoro_clip_amd.video_encoder import VideoEncoder

    torch.manual_seed(0)
    enc = VideoEncoder(num_frames=4, spatial_size=(64, 64), pretrained=False, freeze_ratio=1.0, dropout=0.0,
                       compute_dtype=torch.float32, stochastic_depth_prob=0.0).cuda().eval()
    g = torch.Generator().manual_seed(1)
    frames = torch.randint(0, 256, (2, 2, 4, 64, 64, 3), generator=g, dtype=torch.uint8)
    mean, std = [110.0, 100.0, 95.0], [60.0, 58.0, 62.0]
    with pytest.raises(ValueError):
        enc(frames.cuda())
    enc.set_input_normalization(mean, std)
    with torch.no_grad():
        a = enc(frames.cuda())
        b = enc(((frames.float() - torch.tensor(mean)) / torch.tensor(std)).cuda())
    assert a.shape == (2, 512) and (a - b).abs().max() < 1e-4 * b.abs().max()
    # training through the uint8 path: gradients reach the patch-embedding weights
    enc.train()
    enc(frames.cuda()).sum().backward()
    assert enc.model.conv_proj.weight.grad is not None and torch.isfinite(enc.model.conv_proj.weight.grad).all()


def test_text_length_bucketing_leaves_cls_embedding_unchanged():
    """A batch padded to 512 tokens (dataloaders/video_clip_dataset.py:638-644) and the same batch trimmed to its longest
    report give the same text embedding: padded keys are masked, positions are unchanged."""
    from deepcoro_clip_amd.collate import trim_padding
    from deepcoro_clip_amd.text_encoder import TextEncoder

    torch.manual_seed(0)
    te = TextEncoder(dropout=0.0, freeze_ratio=0.0, compute_dtype=torch.float32,
                     bert_config=dict(vocab_size=1000, num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).cuda().eval()
    g = torch.Generator().manual_seed(2)
    B, S = 4, 512
    lens = [17, 40, 5, 33]
    ids = torch.zeros(B, S, dtype=torch.int64)
    mask = torch.zeros(B, S, dtype=torch.int64)
    for i, n in enumerate(lens):
        ids[i, :n] = torch.randint(1, 1000, (n,), generator=g)
        mask[i, :n] = 1
    tid, tmask = trim_padding(ids, mask, bucket=16)
    assert tid.shape == (B, 48)
    with torch.no_grad():
        full = te(ids.cuda(), mask.cuda())
        short = te(tid.cuda(), tmask.cuda())
    assert (full - short).abs().max() < 2e-5 * full.abs().max().clamp_min(1.0)
