"""MaskFormer head with the SDME mask-embedding block, registry type 'MaskFormerHead'.

One class covers the reference's two layers: mmdet `MaskFormerHead.forward`
(mmdet/models/dense_heads/maskformer_head.py:31-160, 498-586) and the mmseg wrapper's `loss` / `predict` / constructor
(mmseg/models/decode_heads/maskformer_head.py:22-51, 108-180).  The Hungarian-matched loss lives
in loss.py; the benchmark's headline step keeps the data-independent loss of SURVEY 8d.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .head_layers import MLP, DetrTransformerDecoder, SinePositionalEncoding
from .conv import Conv1d, Conv2d, spikes_in
from .fused import bn_act
from . import ops
from .loss import MaskFormerLoss, seg_to_instances
from .neuron import Q_IFNode, Quant
from .registry import MODELS, ConfigDict


# key / value neurons of the decoder straight from  memory (+ level_embed) (+ pos)  by one fused add + neuron kernel
FUSED_KV_NEURONS = True
# the pixel decoder's mask_feature 1x1 convolution folded into the mask contraction (ops.mask_einsum_folded)
FOLD_MASK_FEATURE = True
PREDICT_LAST_ONLY = True          # predict(): SDME block and mask contraction for the last decoder layer only (False: all L + 1, as `forward`)
QUERY_STREAM_CHANNEL_MAJOR = True     # decoder queries channel-major between the layers (no transposes around the projections)


def _lif():
    return Q_IFNode(surrogate_function=Quant())


@MODELS.register_module()
class MaskFormerHead(nn.Module):
    def __init__(self, in_channels, feat_channels, out_channels, num_classes=None, num_things_classes=80,
                 num_stuff_classes=53, num_queries=100, T=4, pixel_decoder=None, enforce_decoder_input_project=False,
                 transformer_decoder=None, positional_encoding=dict(num_feats=128, normalize=True), loss_cls=None,
                 loss_mask=None, loss_dice=None, train_cfg=None, test_cfg=None, init_cfg=None, align_corners=False,
                 ignore_index=255, in_index=None, **kwargs):
        super().__init__()
        if num_classes is None:
            num_classes = num_things_classes + num_stuff_classes
        self.num_things_classes = num_things_classes
        self.num_stuff_classes = num_stuff_classes
        self.num_classes = num_classes
        self.num_queries = num_queries
        transformer_decoder = ConfigDict(transformer_decoder)
        self.num_transformer_decoder_layers = transformer_decoder.num_layers
        self.num_transformer_feat_level = 3
        self.T = T
        self.alpha = 4
        self.Identity = nn.Identity()
        pixel_decoder = ConfigDict(pixel_decoder)
        pixel_decoder.update(in_channels=in_channels, feat_channels=feat_channels, out_channels=out_channels)
        self.pixel_decoder = MODELS.build(pixel_decoder)
        self.transformer_decoder = DetrTransformerDecoder(**transformer_decoder)
        self.decoder_embed_dims = self.transformer_decoder.embed_dims
        self.decoder_input_projs = nn.ModuleList()
        for _ in range(self.num_transformer_feat_level):
            if self.decoder_embed_dims != feat_channels or enforce_decoder_input_project:
                self.decoder_input_projs.append(Conv2d(feat_channels, self.decoder_embed_dims, kernel_size=1))
            else:
                self.decoder_input_projs.append(nn.Identity())
        self.decoder_pe = SinePositionalEncoding(**positional_encoding)
        self.query_embed = nn.Embedding(num_queries, out_channels)
        self.query_feat = nn.Embedding(num_queries, out_channels)
        self.level_embed = nn.Embedding(self.num_transformer_feat_level, feat_channels)
        self.decoder_post_norm = nn.Sigmoid()
        self.decoder_out_spike = _lif()
        self.cls_embed = nn.Linear(feat_channels, num_classes + 1)
        self.mask_embed_spike = _lif()
        self.mask_embed = MLP(in_dim=feat_channels, out_dim=out_channels, layer=3, T=T, quant_const=self.alpha)
        self.w = nn.Parameter(torch.ones(1))
        self.shortcut_conv_spike = _lif()
        self.shortcut_conv = nn.Sequential(Conv1d(num_queries, num_queries, kernel_size=1, stride=1, bias=False),
                                           nn.BatchNorm1d(num_queries))
        spikes_in(self.shortcut_conv[0])                    # reads alpha * spikes = multiples of 1/2
        self.test_cfg, self.train_cfg = test_cfg, train_cfg
        self.criterion = MaskFormerLoss(num_classes, num_queries, loss_cls, loss_mask, loss_dice, train_cfg)
        self.align_corners = align_corners
        self.out_channels = num_classes
        self.ignore_index = ignore_index
        self._pe_cache = {}

    def init_weights(self):
        pass

    def _pos(self, bs, h, w, device):
        """Sine position embedding of an all-valid map, channel-major [bs, C, h*w] (data independent -> cached)."""
        key = (bs, h, w, str(device))
        if key not in self._pe_cache:
            m = torch.zeros((bs, h, w), dtype=torch.bool, device=device)
            self._pe_cache[key] = self.decoder_pe(m).flatten(2).contiguous()
        return self._pe_cache[key]

    def forward(self, x, batch_data_samples=None, last_only=False):
        """x: the 4 backbone maps.  -> all_cls_scores [L+1,B,Q,K+1], all_mask_preds [L+1,B,Q,H/2,W/2].  `last_only` (inference): the
        predictions of the last decoder layer alone ([1, ...]) -- the only ones `predict` reads; the decoder itself runs in full."""
        mask_features, memory, msm = self.pixel_decoder(x, None, spike_memory=True, fold_mask_feature=FOLD_MASK_FEATURE)
        t, bs = memory.shape[:2]
        query_feat = self.query_feat.weight.unsqueeze(0).repeat((t, bs, 1, 1))
        query_embed = self.query_embed.weight.unsqueeze(0).repeat(bs, 1, 1)
        # Keys / values stay in the channel-major layout the pixel decoder produced ([t, b, C, h*w]); the reference
        # transposes them to token-major and back around every projection (maskformer_head.py:535-540).  key + key_pos
        # depends only on the level, so it is formed once per level instead of once per decoder layer.
        dec_in, dec_key, kv_spikes = self.decoder_inputs(msm, bs)
        out_dec = self.run_decoder(query_feat, query_embed, dec_in, dec_key, kv_spikes)
        return self.sdme(out_dec[-1:] if last_only else out_dec, mask_features)

    def decoder_inputs(self, msm, bs):
        """The three memory levels as the decoder layers read them -> (dec_in, dec_key, kv_spikes), one entry per level: the value
        map  msm + level_embed  and the key map  ... + pos  channel-major, or -- fused -- None, None and the (key, value) spike
        maps of the cross-attention's k / v neurons computed straight from the pixel decoder's map (ops.sum2_lif)."""
        nl = self.num_transformer_feat_level
        layers = self.transformer_decoder.layers

        def pure(m):          # reset, stateless, unrecorded: the neuron is a pure function of its input
            return isinstance(m.v, float) and not m.keep_membrane and m.stats is None and not m._forward_hooks

        dec_in, dec_key, kv_spikes = [], [], []
        for i in range(nl):
            d = self.decoder_input_projs[i](msm[i]).flatten(3)
            pos = self._pos(bs, msm[i].shape[-2], msm[i].shape[-1], d.device)
            users = [layers[j].cross_attn.attn for j in range(i, self.num_transformer_decoder_layers, nl)]
            if (FUSED_KV_NEURONS and d.is_cuda and d.shape[-1] % 4 == 0 and users
                    and all(pure(a.k_conv_spike) and pure(a.v_conv_spike) for a in users)
                    and len({(a.k_conv_spike.D, a.k_conv_spike.v_threshold, a.v_conv_spike.D, a.v_conv_spike.v_threshold)
                             for a in users}) == 1 and users[0].k_conv_spike.D == users[0].v_conv_spike.D):
                # the key / value neurons of every layer on this level, straight from the pixel decoder's map: neither
                # d + level_embed nor d + level_embed + pos is materialised (only these neurons read them)
                n0 = users[0].k_conv_spike
                yk, yv = ops.sum2_lif(d.flatten(0, 1), self.level_embed.weight[i], pos, bs, n0.D, n0.v_threshold)
                kv_spikes.append((yk.view(d.shape), yv.view(d.shape)))
                dec_in.append(None); dec_key.append(None)
                continue
            d = d + self.level_embed.weight[i].view(1, 1, -1, 1)
            dec_in.append(d)
            dec_key.append(d + pos)
            kv_spikes.append(None)
        return dec_in, dec_key, kv_spikes

    def run_decoder(self, query_feat, query_embed, dec_in, dec_key, kv_spikes):
        """The transformer decoder (detr_layers.py:491-559 per layer; level cycling maskformer_head.py:554-564) -> the stacked
        [L + 1, T, B, Q, C] queries (initial + after every layer)."""
        nl = self.num_transformer_feat_level
        layers = self.transformer_decoder.layers
        # Keys / values do not depend on the query: with ops.LONG_STREAMS set, the key / value chains of ALL layers (the 1 024 -
        # 16 384-token projections) are launched on a side stream ahead of the serial 100-query chain and overlap with it.
        kv_proj = [None] * self.num_transformer_decoder_layers
        # (a later layer on the same level reads the level's key / value spikes through their spare handles: the neurons' backward kernel
        # sums the two layers' gradients, ops.Spikes.second)
        kv_of = [kv_spikes[i % nl] if (i < nl or kv_spikes[i % nl] is None) else tuple(s.second() for s in kv_spikes[i % nl])
                 for i in range(self.num_transformer_decoder_layers)]
        if ops.LONG_STREAMS and "kv" in ops.LONG_WHAT:
            for i in range(self.num_transformer_decoder_layers):
                lv = i % nl
                attn = layers[i].cross_attn.attn
                kvs = kv_of[i]
                (k, v), handle = ops.fork(
                    1, lambda: attn.project_kv(dec_key[lv], dec_in[lv], True, kvs),
                    inputs=(dec_key[lv], dec_in[lv], kvs), what="kv")
                kv_proj[i] = (k, v, handle)
        out_dec = [query_feat]
        if QUERY_STREAM_CHANNEL_MAJOR and query_feat.is_cuda:
            # the queries travel channel-major between the layers (head_layers.DetrTransformerDecoderLayer.forward_stream)
            q_cm = query_feat.transpose(2, 3).contiguous()
            pos_cm = query_embed.transpose(1, 2).contiguous()
            n = self.num_transformer_decoder_layers
            pos_of = ops.fan_out(pos_cm, 2 * n)          # one alias per reader (two per layer): their gradients are summed by one launch
            for i in range(n):
                lv = i % nl
                q_tm, q_cm = layers[i].forward_stream(q_cm, (pos_of[2 * i], pos_of[2 * i + 1]), key=dec_key[lv], value=dec_in[lv], kv_spikes=kv_of[i],
                                                      kv_projected=kv_proj[i], last=i == n - 1)
                out_dec.append(q_tm)
        else:
            for i in range(self.num_transformer_decoder_layers):
                lv = i % nl
                query_feat = layers[i](
                    query=query_feat, key=dec_key[lv], value=dec_in[lv], query_pos=query_embed, key_pos=None,
                    cross_attn_mask=None, key_padding_mask=None, kv_channel_major=True, kv_spikes=kv_of[i],
                    kv_projected=kv_proj[i])
                out_dec.append(query_feat)
        return torch.stack(out_dec)

    def sdme(self, out_dec, mask_features):
        """The SDME block and the mask contraction (maskformer_head.py:568-586): out_dec [L + 1, T, B, Q, C], mask_features the fp32
        map [T, B, C, H/2, W/2] or mask_feature_spike's bf16 spike map (ops.Spikes: the 1x1 convolution is then folded into the
        contraction) -> (all_cls_scores, all_mask_preds)."""
        ln, t, bs, nq, C = out_dec.shape
        z = self.decoder_post_norm(out_dec)
        a = self.alpha * self.decoder_out_spike(z)
        all_cls_scores = ops.linear_tm(a, self.cls_embed.weight, self.cls_embed.bias).mean(1)
        sc = (self.alpha * self.shortcut_conv_spike(z)).reshape(ln * t * bs, nq, C)
        sc, _ = bn_act(self.shortcut_conv[0].forward_nobias(sc), None, self.shortcut_conv[1])
        e = self.mask_embed(a) + self.w * sc.view(ln, t, bs, nq, C)
        e = self.alpha * self.mask_embed_spike(e)
        # einsum('ltbqc,tbchw->ltbqhw').mean(t) with the mean folded into the contraction: the reference's
        # [L+1,T,B,Q,H,W] intermediate (734 MB / image at 512^2, T=4) is never materialised.
        Hm, Wm = mask_features.shape[-2:]
        eq = e.permute(1, 2, 0, 3, 4).reshape(t, bs, ln * nq, C)        # [t, b, L*Q, C]: all L+1 predictions in one GEMM
        # e = alpha * spikes with alpha = 4: multiples of 1/2 <= 4, exact in bf16 -> the matrix-core path of ops.mask_einsum
        ops.join(getattr(self.pixel_decoder, "mask_feature_handle", None), (mask_features,))   # its side-stream branch
        if isinstance(mask_features, ops.Spikes):
            # the pixel decoder handed over mask_feature_spike's output: the 1x1 mask_feature convolution is folded into the
            # contraction, sum_t (E_t W) @ S_t + bias term -- the convolution and its 537 MB output never run / exist
            mfc = self.pixel_decoder.mask_feature
            acc = ops.mask_einsum_folded(eq, mask_features.flatten(0, 1).flatten(2), mfc.weight.view(mfc.out_channels, -1), mfc.bias,
                                         1.0 / t, t, bs, e_exact=float(self.alpha) == 4.0)
        else:
            acc = ops.mask_einsum(eq, mask_features.flatten(3), 1.0 / t, e_exact=float(self.alpha) == 4.0)      # [b, L*Q, HW]
        all_mask_preds = acc.view(bs, ln, nq, Hm, Wm).permute(1, 0, 2, 3, 4)
        return all_cls_scores, all_mask_preds

    def predict(self, x, batch_img_metas, test_cfg=None):
        """mmseg MaskFormerHead.predict (decode_heads/maskformer_head.py:138-180) -> seg logits [B,K,H,W]."""
        for metainfo in batch_img_metas:                     # as the reference: the patch shape becomes the batch input shape
            metainfo["batch_input_shape"] = metainfo["img_shape"]
        # The segmentation logits depend on the LAST layer's class scores and masks only (cls[-1], masks[-1] below).  In eval mode every
        # op of the SDME block is per sample (BatchNorm on its running statistics), so the other L layers' predictions -- 6/7 of the
        # mask contraction, 0.5 ms of the C2 inference step -- are dead outputs and are not computed.  (Train-mode BatchNorm statistics
        # run over all layers' rows: there the block is evaluated in full.)
        last = PREDICT_LAST_ONLY and not getattr(self, "training", True)
        cls, masks = self(x, None, last_only=True) if last else self(x, None)
        img_shape = batch_img_metas[0]["batch_input_shape"]
        mp = ops.upsample_bilinear(masks[-1], tuple(img_shape), sigmoid=True) if masks.is_cuda else \
            F.interpolate(masks[-1], size=tuple(img_shape), mode="bilinear", align_corners=False).sigmoid()
        cls_score = F.softmax(cls[-1], dim=-1)[..., :-1]
        return ops.class_mask_product(cls_score.contiguous(), mp)

    def loss_by_feat(self, all_cls_scores, all_mask_preds, batch_gt_instances, batch_img_metas=None):
        """mmdet MaskFormerHead.loss_by_feat (dense_heads/maskformer_head.py:376-414); `batch_gt_instances`: per image an
        object with `.labels` / `.masks` or a (labels, masks) pair."""
        gts = [g if isinstance(g, (tuple, list)) else (g.labels, g.masks) for g in batch_gt_instances]
        return self.criterion.loss_by_feat(all_cls_scores, all_mask_preds, gts, reduce_fn=self._reduce_fn())

    @staticmethod
    def _reduce_fn():
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            def reduce_fn(t):                                  # reduce_mean of num_total_masks (:459), all layers at once
                t = t.clone()
                torch.distributed.all_reduce(t.div_(torch.distributed.get_world_size()))
                return t
            return reduce_fn
        return None

    def loss(self, x, batch_data_samples, train_cfg=None):
        """mmseg MaskFormerHead.loss (decode_heads/maskformer_head.py:108-136): semantic maps -> per-class binary masks ->
        forward -> Hungarian-matched loss dictionary.  `batch_data_samples`: SegDataSample-like objects
        (`.gt_sem_seg.data` [1,H,W]) or the semantic maps themselves."""
        segs = [d if torch.is_tensor(d) else d.gt_sem_seg.data for d in batch_data_samples]
        all_cls_scores, all_mask_preds = self(x, batch_data_samples)
        if len({tuple(s.shape[-2:]) for s in segs}) == 1:
            seg_all = torch.stack([s.reshape(s.shape[-2:]) for s in segs])
            if self.criterion.semantic_ok(all_mask_preds, seg_all):
                # semantic maps at twice the predictions' resolution on the GPU (every Spike2Former config): static-shape device
                # side, no gathers (loss.MaskFormerLoss.loss_semantic) -- same dictionary as the generic path below
                return self.criterion.loss_semantic(all_cls_scores, all_mask_preds, seg_all, self.ignore_index, self._reduce_fn())
        gts = [seg_to_instances(seg, self.ignore_index) for seg in segs]
        return self.loss_by_feat(all_cls_scores, all_mask_preds, gts)
