"""The ``state_dict`` contract of ``SceneDiffusionModel``.

Reference: model/sdm.py:47-126 builds the modules whose parameter names/shapes are listed here;
run/test_sdm.py:124 loads checkpoints with ``strict=True`` so the names are part of the drop-in
boundary.  The table is data, generated from the layer dimensions below; nothing here computes.
"""
from collections import OrderedDict

SA_MLPS = (  # model/pcd_backbone/pointnet2.py:46-49  (in_channel, [mlp])
    ("sa1", 6, (32, 32, 64)),
    ("sa2", 64 + 3, (64, 64, 128)),
    ("sa3", 128 + 3, (128, 128, 256)),
    ("sa4", 256 + 3, (256, 256, 512)),
)
FP_MLPS = (  # pointnet2.py:50-55
    ("fp4", 768, (256, 256)),
    ("fp3", 384, (256, 256)),
    ("fp2", 320, (256, 128)),
    ("fp1", 128, (128, 128, 128)),
)


def _linear(spec, name, out_f, in_f, kind="linear"):
    spec[name + ".weight"] = ((out_f, in_f), kind + "_w")
    spec[name + ".bias"] = ((out_f,), "bias")


def _bn(spec, name, c):
    spec[name + ".weight"] = ((c,), "norm_w")
    spec[name + ".bias"] = ((c,), "norm_b")
    spec[name + ".running_mean"] = ((c,), "bn_mean")
    spec[name + ".running_var"] = ((c,), "bn_var")
    spec[name + ".num_batches_tracked"] = ((), "bn_count")


def state_dict_spec(max_cats=13, clip_dim=512, latent_dim=128, cat_emb=32, pcd_points=1024, pcd_dim=3,
                    xyz_dim=3, translation_params=12, max_len=5000):
    """Ordered {name: (shape, kind)} in the reference's registration order."""
    s = OrderedDict()
    L = latent_dim
    _linear(s, "embed_text.0", clip_dim // 2, clip_dim)
    _linear(s, "embed_text.2", 2 * L, clip_dim // 2)
    _linear(s, "embed_text.4", L, 2 * L)
    s["sequence_pos_encoder.pe"] = ((max_len, 1, L), "pe")
    s["embed_timestep.sequence_pos_encoder.pe"] = ((max_len, 1, L), "pe")
    _linear(s, "embed_timestep.time_embed.0", L, L)
    _linear(s, "embed_timestep.time_embed.2", L, L)
    _linear(s, "embed_cat.0", cat_emb, max_cats)
    _linear(s, "predict_cat.0", L // 2, L)
    _linear(s, "predict_cat.2", L // 4, L // 2)
    _linear(s, "predict_cat.4", max_cats, L // 4)
    # nn.MultiheadAttention(embed_dim=L, kdim=cat_emb, vdim=pcd_points*pcd_dim)  (sdm.py:79)
    s["attn_layer.q_proj_weight"] = ((L, L), "linear_w")
    s["attn_layer.k_proj_weight"] = ((L, cat_emb), "linear_w")
    s["attn_layer.v_proj_weight"] = ((L, pcd_points * pcd_dim), "linear_w")
    s["attn_layer.in_proj_bias"] = ((3 * L,), "bias")
    _linear(s, "attn_layer.out_proj", L, L)
    _linear(s, "translation_layer.0", L, L + cat_emb)
    _linear(s, "translation_layer.2", translation_params, L)
    _linear(s, "point_wise_trans_layer.0", xyz_dim, translation_params + xyz_dim)
    # nn.MultiheadAttention(embed_dim=12, heads=12, kdim=vdim=3)  (sdm.py:95)
    T = translation_params
    s["pcd_attention.q_proj_weight"] = ((T, T), "linear_w")
    s["pcd_attention.k_proj_weight"] = ((T, xyz_dim), "linear_w")
    s["pcd_attention.v_proj_weight"] = ((T, xyz_dim), "linear_w")
    s["pcd_attention.in_proj_bias"] = ((3 * T,), "bias")
    _linear(s, "pcd_attention.out_proj", T, T)
    for name, cin, mlp in SA_MLPS:
        last = cin
        for i, c in enumerate(mlp):
            s["pcd_backbone.%s.mlp_convs.%d.weight" % (name, i)] = ((c, last, 1, 1), "linear_w")
            s["pcd_backbone.%s.mlp_convs.%d.bias" % (name, i)] = ((c,), "bias")
            last = c
        for i, c in enumerate(mlp):
            _bn(s, "pcd_backbone.%s.mlp_bns.%d" % (name, i), c)
    for name, cin, mlp in FP_MLPS:
        last = cin
        for i, c in enumerate(mlp):
            s["pcd_backbone.%s.mlp_convs.%d.weight" % (name, i)] = ((c, last, 1), "linear_w")
            s["pcd_backbone.%s.mlp_convs.%d.bias" % (name, i)] = ((c,), "bias")
            last = c
        for i, c in enumerate(mlp):
            _bn(s, "pcd_backbone.%s.mlp_bns.%d" % (name, i), c)
    s["pcd_backbone.conv1.weight"] = ((128, 128, 1), "linear_w")
    s["pcd_backbone.conv1.bias"] = ((128,), "bias")
    _bn(s, "pcd_backbone.bn1", 128)
    s["pcd_backbone.conv2.weight"] = ((pcd_dim, 128, 1), "linear_w")
    s["pcd_backbone.conv2.bias"] = ((pcd_dim,), "bias")
    for i, (co, ci) in enumerate(((64, xyz_dim), (64, 64), (64, 64))):
        _linear(s, "human_backbone.de_spiral.%d.conv.layer" % i, co, ci)
        s["human_backbone.de_spiral.%d.norm.weight" % i] = ((co,), "norm_w")
        s["human_backbone.de_spiral.%d.norm.bias" % i] = ((co,), "norm_b")
    _linear(s, "human_backbone.de_spiral.3.layer", 3, 64)
    _linear(s, "upsampling_layer.0", 128, 1)
    _linear(s, "upsampling_layer.2", 512, 128)
    _linear(s, "upsampling_layer.4", pcd_points, 512)
    _linear(s, "combine_extraction.0", L, 2 * L)
    _linear(s, "input_process.pose_embedding.0", L // 2, xyz_dim)
    _linear(s, "input_process.pose_embedding.2", L, L // 2)
    _linear(s, "input_process.combination_extraction.0", int(L * 1.5), 2 * L)
    _linear(s, "input_process.combination_extraction.2", L, int(L * 1.5))
    _linear(s, "output_process.pose_final.0", L // 2, L)
    _linear(s, "output_process.pose_final.2", xyz_dim, L // 2)
    return s
