"""MI355X-native GIF discriminator — drop-in for /root/reference/model/stg2_discriminator.py:8-76.

Same constructor / forward() signature, attribute names and state_dict keys (convs.*, final_conv.*,
final_linear.*).  Twice differentiable w.r.t. `input` (R1, train.py:148): every layer is built from the
any-order autograd Functions of gif_amd.functional.  Internally NHWC with the 9-channel image+condition
input zero-padded to 12 channels and the 513-channel stddev tensor to 516.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from . import functional as GF
from .layers import ConvLayer, EqualLinear, ResBlock
from .ops import cpad


class Discriminator(nn.Module):
    def __init__(self, size, channel_multiplier=2, num_color_chnls=3, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        cm = channel_multiplier
        channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm,
                    1024: 16 * cm}
        convs = [ConvLayer(num_color_chnls, channels[size], 1)]
        log_size = int(math.log(size, 2))
        in_channel = channels[size]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            convs.append(ResBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.convs = nn.Sequential(*convs)
        self.stddev_group = 4
        self.stddev_feat = 1
        # > 1: the batch is that many independent calls stacked along dim 0 (GifTrainer runs D on [real; fake] in one pass):
        # the minibatch-stddev statistic — the only cross-sample coupling — is taken inside each chunk, exactly as in separate calls
        self.stddev_chunks = 1
        self.act_dtype = torch.float32  # see StyledGenerator.act_dtype
        self.final_conv = ConvLayer(in_channel + 1, channels[4], 3)
        self.final_linear = nn.Sequential(
            EqualLinear(channels[4] * 4 * 4, channels[4], activation='fused_lrelu'),
            EqualLinear(channels[4], 1),
        )

    def set_activation_dtype(self, dtype):
        if dtype not in (torch.float32, torch.float16):
            raise ValueError(f"activation dtype must be torch.float32 or torch.float16, got {dtype}")
        self.act_dtype = dtype
        return self

    def forward(self, input, condition=None, step=0, alpha=0):
        if type(input) in (list, tuple):
            input = input[0]
        # torch.cat((input, condition), 1) (reference :50-53) + channel padding (9 -> 12 / 16) + conversion to the activation
        # dtype + NHWC layout: one HIP pass (round 3: four ATen passes, 3.3 ms per iteration at 1024^2)
        c = input.shape[1] + (condition.shape[1] if condition is not None else 0)
        out = self.convs(GF.pack_nhwc(input, condition, cpad(c, self.act_dtype), self.act_dtype))
        batch, channel, height, width = out.shape
        # [B,C,4,4] -> [B,cpad(C+1),4,4]: channel C is the group's mean stddev (wavefront-shuffle reduction in HIP)
        if self.stddev_chunks > 1:
            assert batch % self.stddev_chunks == 0, (batch, self.stddev_chunks)
            out = torch.cat([GF.minibatch_stddev(h, min(h.shape[0], self.stddev_group), cpad(channel + 1, out.dtype))
                             for h in out.chunk(self.stddev_chunks, dim=0)], dim=0)
        else:
            out = GF.minibatch_stddev(out, min(batch, self.stddev_group), cpad(channel + 1, out.dtype))
        out = self.final_conv(out)
        out = out.reshape(batch, -1).float()  # logical NCHW order == the reference's view(batch, -1); the head is fp32
        out = self.final_linear(out)
        return out, None
