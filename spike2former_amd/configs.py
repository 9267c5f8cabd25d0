"""The BASELINE.json workloads as model configs, in the reference's own config vocabulary
(configs/Spike2Former/SDTv2_maskformer_DCNpixelDecoder_ade20k.py:23-131)."""
from .registry import ConfigDict

WORKLOADS = {
    # name: H, W, T, per-GPU batch, classes, embed_dim, feat_channels, queries, pd layers/ffn, dec layers/ffn, group, num_feats
    "C1": dict(H=128, W=128, T=1, B=1, K=20, embed_dim=[16, 32, 64, 72], Fc=64, Q=10, pd=(2, 256), dec=(2, 512), G=8, nf=32),
    "C1_64": dict(H=64, W=64, T=2, B=2, K=20, embed_dim=[16, 32, 64, 72], Fc=64, Q=10, pd=(2, 256), dec=(2, 512), G=8, nf=32),
    "C2": dict(H=512, W=512, T=4, B=2, K=150, embed_dim=[64, 128, 256, 360], Fc=256, Q=100, pd=(6, 1024), dec=(6, 2048), G=32, nf=128),
    "C3": dict(H=512, W=1024, T=4, B=2, K=19, embed_dim=[64, 128, 256, 360], Fc=256, Q=100, pd=(6, 2048), dec=(6, 2048), G=32, nf=128),
    # BASELINE configs[4]: COCO-panoptic-shaped, E-SpikeFormer (SDT-v3) backbone; 800x1333 padded to a multiple of 32;
    # 80 things + 53 stuff classes (the head's defaults); no such config ships with the reference: synthesised
    "C5": dict(H=800, W=1344, T=4, B=1, K=133, embed_dim=[64, 128, 256, 360], Fc=256, Q=100, pd=(6, 1024), dec=(6, 2048), G=32, nf=128,
               backbone="Spiking_vit_MetaFormerv2", things=80, stuff=53),
    "C5_tiny": dict(H=64, W=96, T=2, B=1, K=12, embed_dim=[16, 32, 64, 72], Fc=64, Q=10, pd=(2, 256), dec=(2, 512), G=8, nf=32,
                    backbone="Spiking_vit_MetaFormerv2", things=7, stuff=5),
    "C4": dict(H=512, W=512, T=8, B=2, K=150, embed_dim=[64, 128, 256, 360], Fc=256, Q=100, pd=(6, 1024), dec=(6, 2048), G=32, nf=128),
}


def model_cfg(name):
    w = WORKLOADS[name]
    e, Fc = w["embed_dim"], w["Fc"]
    norm_cfg = dict(type="SyncBN", requires_grad=True)       # accepted and ignored, as in the reference
    return ConfigDict(
        type="EncoderDecoder",
        backbone=dict(type=w.get("backbone", "Spiking_vit_MetaFormer"), img_size_h=w["H"], img_size_w=w["W"], patch_size=16,
                      embed_dim=e, num_heads=8, mlp_ratios=4, in_channels=3, num_classes=w["K"], qkv_bias=False, depths=8,
                      sr_ratios=1, T=w["T"], norm_eval=True, norm_cfg=norm_cfg,
                      decode_mode="QTrick" if w.get("backbone") == "Spiking_vit_MetaFormerv2" else "Qsnn"),
        decode_head=dict(
            type="MaskFormerHead", in_channels=[e[0] // 2, e[0], e[1], e[3]], feat_channels=Fc, in_index=[0, 1, 2, 3],
            num_classes=w["K"], out_channels=Fc, num_queries=w["Q"], T=w["T"],
            num_things_classes=w.get("things", w["K"]), num_stuff_classes=w.get("stuff", 0),
            pixel_decoder=dict(
                type="mmdet.DCNTransformerEncoderPixelDecoder", norm_cfg=norm_cfg, T=w["T"],
                encoder=dict(num_layers=w["pd"][0], layer_cfg=dict(
                    self_attn_cfg=dict(embed_dims=Fc, num_heads=8, batch_first=True, dw_kernel_size=5, group=w["G"]),
                    ffn_cfg=dict(embed_dims=Fc, feedforward_channels=w["pd"][1], num_fcs=2))),
                positional_encoding=dict(num_feats=w["nf"], normalize=True)),
            enforce_decoder_input_project=False,
            loss_cls=dict(type="mmdet.CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0, reduction="mean",
                          class_weight=[1.0] * w["K"] + [0.1]),
            loss_mask=dict(type="mmdet.FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=20.0),
            loss_dice=dict(type="mmdet.DiceLoss", use_sigmoid=True, activate=True, reduction="mean", naive_dice=True, eps=1.0,
                           loss_weight=1.0),
            train_cfg=dict(assigner=dict(type="mmdet.HungarianAssigner", match_costs=[
                dict(type="mmdet.ClassificationCost", weight=1.0),
                dict(type="mmdet.FocalLossCost", weight=20.0, binary_input=True),
                dict(type="mmdet.DiceCost", weight=1.0, pred_act=True, eps=1.0)]),
                sampler=dict(type="mmdet.MaskPseudoSampler")),
            positional_encoding=dict(num_feats=w["nf"], normalize=True),
            transformer_decoder=dict(
                return_intermediate=True, num_layers=w["dec"][0],
                layer_cfg=dict(
                    self_attn_cfg=dict(embed_dims=Fc, num_heads=8, attn_type="SA", batch_first=True),
                    cross_attn_cfg=dict(embed_dims=Fc, num_heads=8, attn_type="CA", batch_first=True),
                    ffn_cfg=dict(embed_dims=Fc, feedforward_channels=w["dec"][1], num_fcs=2, add_identity=True)),
                init_cfg=None)),
        train_cfg=dict(), test_cfg=dict(mode="whole"))
