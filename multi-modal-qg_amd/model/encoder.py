"""``model.encoder`` of the reference, re-implemented on the HIP kernels.

Follows /root/reference/model/encoder.py: TextEncoder (:80-111), VideoConvLstmEncoder (:31-78),
AudioVideoEncoder (:113-131), AudioEncoder (:8-19), VideoResnetEncoder (:21-29) — same class
names, constructor / ``forward`` signatures and state-dict keys.

* LSTMs and the embedding lookup run in the HIP kernels (no CPU path).
* The four conv+ReLU+BatchNorm(+max-pool) blocks of the frame encoder run in the HIP frame-CNN
  kernels (``mmqg_frame_cnn_fwd/bwd``) for the reference's 3x3 / stride-1 hyper-parameters; its
  LSTM stage uses the HIP sequence executor.  ``VideoConvLstmEncoder.forward`` also accepts pre-extracted per-frame features (T,D) / (B,T,D) that skip the CNN
  (BASELINE configs 2-5: ``video_emb_dim`` = feature width, e.g. 2048).
* AudioEncoder wraps a remote torch.hub VGGish in the reference (encoder.py:12), which cannot
  be fetched offline; here it passes (n_clips,128) feature tensors through unchanged.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from ._params import LSTMParams, fresh_seed


class AudioEncoder(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, audio_file):
        if not torch.is_tensor(audio_file):
            raise NotImplementedError(
                "AudioEncoder: the VGGish front-end (torch.hub 'harritaylor/torchvggish', reference "
                "encoder.py:12) is outside this build; pass pre-extracted (n_clips,128) features")
        return audio_file


class VideoResnetEncoder(nn.Module):
    def __init__(self, download_pretrained=False):
        super().__init__()
        raise NotImplementedError("VideoResnetEncoder (torchvision r2plus1d_18, reference encoder.py:21-29) is never "
                                  "used by train.py and is outside this build")

    def forward(self, video_frames):  # pragma: no cover
        raise NotImplementedError


def per_question_batchnorm(x, bn: nn.BatchNorm2d, valid, training: bool):
    """BatchNorm2d whose statistics span the frames of ONE question (encoder.py:64 feeds
    (T,C,H,W)), for a batch of questions.  x (B,T,C,h,w), valid (B,T) bool.  Running statistics
    advance once per question in batch order, as B sequential reference calls would."""
    B, T, Cc, hh, ww = x.shape
    if not training:
        mean = bn.running_mean.view(1, 1, -1, 1, 1)
        var = bn.running_var.view(1, 1, -1, 1, 1)
    else:
        m = valid.view(B, T, 1, 1, 1).to(x.dtype)
        n = (valid.sum(dim=1).to(x.dtype) * hh * ww).view(B, 1, 1, 1, 1)
        mean = (x * m).sum(dim=(1, 3, 4), keepdim=True) / n
        var = (((x - mean) ** 2) * m).sum(dim=(1, 3, 4), keepdim=True) / n
        with torch.no_grad():
            mom = bn.momentum
            nb = n.view(B, 1)
            mb, vb = mean.view(B, Cc), var.view(B, Cc) * nb / (nb - 1)
            # r <- (1-m)^B r + sum_b m (1-m)^(B-1-b) stat_b
            w = mom * (1 - mom) ** torch.arange(B - 1, -1, -1, device=x.device, dtype=x.dtype)
            bn.running_mean.mul_((1 - mom) ** B).add_((w.view(B, 1) * mb).sum(0))
            bn.running_var.mul_((1 - mom) ** B).add_((w.view(B, 1) * vb).sum(0))
            bn.num_batches_tracked += B
    y = (x - mean) / torch.sqrt(var + bn.eps)
    return y * bn.weight.view(1, 1, -1, 1, 1) + bn.bias.view(1, 1, -1, 1, 1)


class VideoConvLstmEncoder(nn.Module):
    def __init__(self, in_channels, kernel_sz, stride, hidden_dim, video_emb_dim):
        super().__init__()
        self.in_channels, self.kernel_sz, self.stride = in_channels, kernel_sz, stride
        self.hidden_dim, self.video_emb_dim = hidden_dim, video_emb_dim
        self.conv1 = nn.Conv2d(in_channels, 4, kernel_sz, stride)
        self.bn1 = nn.BatchNorm2d(4)
        self.conv2 = nn.Conv2d(4, 6, kernel_sz, stride)
        self.bn2 = nn.BatchNorm2d(6)
        self.conv3 = nn.Conv2d(6, 8, kernel_sz, stride)
        self.bn3 = nn.BatchNorm2d(8)
        self.conv4 = nn.Conv2d(8, 10, kernel_sz, stride)
        self.bn4 = nn.BatchNorm2d(10)
        self.lstm = LSTMParams(video_emb_dim, hidden_dim, 1)
        self.initialise_weights()

    # -- CNN stage ---------------------------------------------------------------------------
    def cnn_features(self, frames_btchw, n_frames=None):
        """(B,T,C,H,W) frames, already in the layout the reference's ``view`` produces ->
        (B,T,flatten).  ReLU comes BEFORE BatchNorm, as at encoder.py:64-65.  The reference's
        hyper-parameters (3x3 kernels, stride 1: config.py:66-67) run in the HIP frame-CNN kernels;
        any other kernel size / stride takes PyTorch-ROCm's convolution ops."""
        B, T = frames_btchw.shape[:2]
        dev = frames_btchw.device
        ops.require_device(frames_btchw)
        if self.kernel_sz == 3 and self.stride == 1:
            params = []
            for i in (1, 2, 3, 4):
                conv, bn = getattr(self, f"conv{i}"), getattr(self, f"bn{i}")
                params += [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
            bn = self.bn1
            z = ops.FrameCNNFn.apply(frames_btchw, n_frames, self.training, bn.eps, bn.momentum,
                                     (False, True, False, True), *params)
            if self.training:
                with torch.no_grad():
                    for i in (1, 2, 3, 4):
                        getattr(self, f"bn{i}").num_batches_tracked += B
            return z.reshape(B, T, -1)
        valid = (torch.arange(T, device=dev).view(1, -1) < n_frames.view(-1, 1).to(dev)) if n_frames is not None \
            else torch.ones(B, T, dtype=torch.bool, device=dev)
        x = frames_btchw
        for i, pool in ((1, False), (2, True), (3, False), (4, True)):
            conv, bn = getattr(self, f"conv{i}"), getattr(self, f"bn{i}")
            y = F.relu(conv(x.reshape(B * T, *x.shape[2:])))
            y = per_question_batchnorm(y.view(B, T, *y.shape[1:]), bn, valid, self.training)
            if pool:
                z = F.max_pool2d(y.reshape(B * T, *y.shape[2:]), self.kernel_sz, self.kernel_sz)
                y = z.view(B, T, *z.shape[1:])
            x = y
        return x.reshape(B, T, -1)

    def lstm_features(self, feats_tbd, n_frames=None):
        """(T,B,D) features -> (T,B,hidden) through the HIP LSTM executor (zero initial state)."""
        T, B, _ = feats_tbd.shape
        zeros = torch.zeros(1, B, self.hidden_dim, device=feats_tbd.device)
        y, _, _ = ops.lstm_seq(feats_tbd.contiguous(), zeros, zeros.clone(), self.lstm.flat(), 0.0, self.training, 0)
        return y

    def forward(self, video_frames):
        ops.require_device(video_frames)
        if video_frames.dim() == 5:
            if video_frames.shape[0] != 1:
                raise RuntimeError("VideoConvLstmEncoder.forward expects (1,C,T,H,W) like the reference; use "
                                   "cnn_features/lstm_features for batches")
            _, Cc, T, hh, ww = video_frames.shape
            frames = video_frames.contiguous().view(1, T, Cc, hh, ww)       # encoder.py:64: raw view, not a permute
            feats = self.cnn_features(frames).transpose(0, 1)               # (T,1,flatten)
        elif video_frames.dim() == 2:
            feats = video_frames.unsqueeze(1)                               # (T,D) features of one question
        else:
            feats = video_frames.transpose(0, 1)                            # (B,T,D) -> (T,B,D)
        return self.lstm_features(feats)                                    # (T,B,hidden); (T,1,hidden) for one question

    def initialise_weights(self):
        self.lstm.reference_init()


class TextEncoder(nn.Module):
    def __init__(self, num_layers, dropout_p, hidden_dim, emb_dim, emb_layer, device):
        super().__init__()
        self.num_layers, self.hidden_dim, self.embedding_dim = num_layers, hidden_dim, emb_dim
        self.word_embeddings = emb_layer
        self.device = device
        self.dropout_p = dropout_p
        self.lstm = LSTMParams(emb_dim, hidden_dim, num_layers, dropout=dropout_p)
        self.initialise_weights()

    def forward(self, text, hidden):
        h, c = hidden
        ops.require_device(h, c, self.word_embeddings.weight)
        B = h.shape[1]
        text = text.to(h.device)
        if B == 1:
            T = text.numel()                                    # encoder.py:96-98: n tokens of ONE question
        else:
            if text.shape[0] != B:
                raise RuntimeError(f"expected {B} rows of token ids, got {tuple(text.shape)}")
            text = text.reshape(B, -1).t()                      # (T,B) time-major
            T = text.shape[0]
        emb = ops.EmbeddingFn.apply(self.word_embeddings.weight, text.reshape(-1))
        y, h_new, c_new = ops.lstm_seq(emb.view(T, B, -1), h, c, self.lstm.flat(), self.dropout_p, self.training,
                                       fresh_seed() if (self.training and self.dropout_p > 0) else 0)
        return y, (h_new, c_new)

    def initialise_weights(self):
        self.lstm.reference_init()

    def init_state(self, batch_sz):
        dev = self.word_embeddings.weight.device
        return (torch.zeros(self.num_layers, batch_sz, self.hidden_dim, device=dev),
                torch.zeros(self.num_layers, batch_sz, self.hidden_dim, device=dev))


class AudioVideoEncoder(nn.Module):
    def __init__(self, av_in_channels, av_kernel_sz, av_stride, av_hidden_dim, video_emb_dim):
        super().__init__()
        self.audio_enc = AudioEncoder()
        self.video_enc = VideoConvLstmEncoder(av_in_channels, av_kernel_sz, av_stride, av_hidden_dim, video_emb_dim)

    def forward(self, audio_file, video_frames):
        # The reference flattens the audio features to one row (encoder.py:123), which only
        # satisfies the decoder's bmm (decoder.py:95) for a single clip; the decoder's contract
        # is (n_clips, audio_emb_dim) rows, which is what is returned here.
        audio_emb = self.audio_enc(audio_file)
        video_emb = self.video_enc(video_frames).squeeze()
        return audio_emb, video_emb
