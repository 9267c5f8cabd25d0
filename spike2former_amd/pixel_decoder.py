"""DCN pixel decoder, registry type 'mmdet.DCNTransformerEncoderPixelDecoder'
(mmdet/models/layers/pixel_decoder.py:316-472; the base-class neuron it inherits and uses is :80)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .head_layers import DCNDetrTransformerEncoder, SinePositionalEncoding
from . import ops
from .conv import Conv1d, Conv2d, spikes_in
from .fused import bn_act, conv_bn_act
from .neuron import Q_IFNode, Quant
from .registry import MODELS


def _lif():
    return Q_IFNode(surrogate_function=Quant())


@MODELS.register_module()
class DCNTransformerEncoderPixelDecoder(nn.Module):
    """forward(feats[4] of [T,B,C_i,H_i,W_i]) -> (mask_feature [T,B,C,H/2,W/2], memory [T,B,C,H/16,W/16], [y16,y8,y4]).

    Module registration order follows the reference (base `PixelDecoder.__init__` first, then the overrides), so that
    `named_modules()` -- and with it the firing table of cal_firing_num.py -- lists neurons in the same order.
    `encoder_in_proj_spike` is constructed but never called, as in the reference (:396 vs :435).
    """

    def __init__(self, in_channels, feat_channels, out_channels, T=4, norm_cfg=None, act_cfg=None, encoder=None,
                 positional_encoding=dict(num_feats=128, normalize=True), init_cfg=None):
        super().__init__()
        self.in_channels = in_channels
        self.feat_channels = feat_channels
        self.num_inputs = len(in_channels)
        self.use_bias = norm_cfg is None
        self.T = T
        # names in the order the reference's base class registers them (pixel_decoder.py:57-87)
        self.lateral_convs = nn.ModuleList()
        self.lateral_convs_spike = nn.ModuleList()
        self.output_convs = nn.ModuleList()
        self.output_convs_spike = nn.ModuleList()
        for i in range(self.num_inputs - 1):
            self.lateral_convs.append(nn.Sequential(Conv2d(in_channels[i], feat_channels, kernel_size=1, stride=1),
                                                    nn.BatchNorm2d(feat_channels)))
            self.lateral_convs_spike.append(_lif())
            self.output_convs.append(nn.Sequential(
                Conv2d(feat_channels, feat_channels, kernel_size=3, padding=1, groups=feat_channels, bias=False),
                nn.BatchNorm2d(feat_channels)))
            self.output_convs_spike.append(_lif())
        self.last_feat_conv_spike = _lif()
        self.last_feat_conv = None
        self.mask_feature_spike = _lif()
        self.mask_feature = Conv2d(feat_channels, out_channels, kernel_size=1, stride=1)
        self.encoder = DCNDetrTransformerEncoder(**encoder)
        self.encoder_embed_dims = self.encoder.embed_dims
        assert self.encoder_embed_dims == feat_channels, (
            f"embed_dims({feat_channels}) of tranformer encoder must equal to feat_channels({self.encoder_embed_dims})")
        self.positional_encoding = SinePositionalEncoding(**positional_encoding)
        self.encoder_in_proj_spike = _lif()
        self.encoder_in_proj = nn.Sequential(Conv2d(in_channels[-1], feat_channels, kernel_size=1, stride=1),
                                             nn.BatchNorm2d(feat_channels))
        self.encoder_out_proj_spike = _lif()
        self.encoder_out_proj = nn.Sequential(Conv2d(feat_channels, feat_channels, kernel_size=1, stride=1),
                                              nn.BatchNorm2d(feat_channels))
        spikes_in(self.mask_feature, self.encoder_in_proj[0], self.encoder_out_proj[0], *[s[0] for s in self.lateral_convs])

    def init_weights(self):
        pass

    def forward(self, feats, batch_img_metas=None, spike_memory=False, fold_mask_feature=False):
        """`spike_memory`: hand `memory` (a spike map) out as the ops.Spikes pair instead of converting it to the fp32 tensor
        of the reference's interface (the head only reads its shape).
        `fold_mask_feature`: return mask_feature_spike's OUTPUT (the bf16 spike map [T, B, C, H/2, W/2]) in place of
        mask_feature(...) when that is possible: the head then folds the 1x1 convolution into its mask contraction
        (ops.mask_einsum_folded) and the 537 MB mask_features tensor never exists."""
        x4 = feats[-1]
        t, bs, c, h, w = x4.shape
        E = self.encoder_embed_dims
        def conv_bn(seq, x, **kw):
            return conv_bn_act(seq[0], x, seq[1], **kw)

        # The lateral 1x1 convolutions read only the backbone taps: with ops.LONG_STREAMS set they are launched on a side
        # stream now and overlap with the six encoder layers (chains of 32x32-map kernels that leave most CUs idle).
        def lateral(i):
            def run():
                x = self.lateral_convs_spike[i].fire(feats[i]).flatten(0, 1)
                ops.use_here(x)
                return self.lateral_convs[i][0].forward_nobias(x)
            return run
        # (inference: the lateral convolution, its BatchNorm, the top-down add and the output neuron are ONE launch below --
        # conv_bn_act's eval fusion -- so nothing is started ahead)
        lat_eval = (not self.training) and not torch.is_grad_enabled()
        lat = {} if lat_eval else {i: ops.fork(0, lateral(i), inputs=(feats[i],), what="lat")
                                   for i in range(self.num_inputs - 2, -1, -1)}

        y = self.last_feat_conv_spike.fire(x4)
        y = conv_bn(self.encoder_in_proj, y.flatten(0, 1))[0].reshape(t, bs, E, h, w)
        memory = self.encoder.forward_nchw(y)          # == encoder(query=y.permute(0,1,3,4,2)).permute(0,1,4,2,3)
        memory = self.encoder_out_proj_spike.fire(memory)
        y = conv_bn(self.encoder_out_proj, memory.flatten(0, 1))[0]
        out = []

        # A level map y has two readers: the next (finer) level's up-sampling and the transformer decoder (`out`).  The decoder reads
        # the pass-through the up-sampling hands back (ops.upsample_bilinear skip=True): its gradient is summed inside the up-sampling's
        # adjoint kernel instead of by an add launch of the autograd engine over the [T*B, C, H, W] map.
        def level(i, y, keep=None):
            up, through = ops.upsample_bilinear(y, feats[i].shape[-2:], skip=True)
            if keep is not None:
                keep.append(through.reshape(t, bs, *through.shape[1:]))
            if lat_eval:
                x = self.lateral_convs_spike[i].fire(feats[i]).flatten(0, 1)
                _, s = conv_bn_act(self.lateral_convs[i][0], x, self.lateral_convs[i][1], residual=up, lif=self.output_convs_spike[i])
            else:
                z, handle = lat[i]
                ops.join(handle, (z,))
                # cur + upsample(y), then the output neuron: residual add and neuron fused into the BatchNorm kernel
                _, s = bn_act(z, self.lateral_convs[i][0].bias, self.lateral_convs[i][1], residual=up, lif=self.output_convs_spike[i])
            # the last level feeds only mask_feature_spike: that neuron is applied by the same BatchNorm kernel (prefire)
            # instead of a separate pass over the [T*B, C, H/2, W/2] map (537 MB at C2)
            return conv_bn(self.output_convs[i], s, next_lif=self.mask_feature_spike if i == 0 else None)[0]

        for i in range(self.num_inputs - 2, 0, -1):
            y = level(i, y, out)          # (appends the map that entered this level)

        # The H/2 level and the mask_feature convolution (the largest maps of the head) feed only the mask einsum at the very
        # end: as a branch on the side stream they overlap with the transformer decoder; the head joins it (mask_feature_handle).
        mfc = self.mask_feature
        fold = (fold_mask_feature and ops.SPIKES_BF16 and mfc.kernel_size == (1, 1) and mfc.stride == (1, 1) and mfc.groups == 1
                and mfc.in_channels % 32 == 0 and not self.mask_feature_spike._forward_hooks)

        def finest(y=y):
            y0 = level(0, y, out)          # (the pass-through is a view of y: nothing of it is computed on the side stream)
            s0 = self.mask_feature_spike.fire(y0)
            if fold and isinstance(s0, ops.Spikes) and s0.tok is not None and (s0.shape[-1] * s0.shape[-2]) % 8 == 0:
                return s0
            return self.mask_feature(s0)
        mf, self.mask_feature_handle = ops.fork(0, finest, inputs=(y,), what="mf")
        return mf.reshape(t, bs, *mf.shape[1:]), (memory if spike_memory else memory.float()), out[:3]
