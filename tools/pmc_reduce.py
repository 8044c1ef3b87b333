"""Reduce the PMC passes of tools/pmc_traffic.sh to profiles/r0N_pmc_traffic.json (bytes per launch and kernel).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte requests of wide coalesced reads
as 64 bytes -> raw read KB doubled; WRITE_SIZE taken as is.  Counter values are KB per dispatch."""
import collections
import csv
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
GROUPS = (("vlfuse_i2t", "vlfuse_i2t_kernel"), ("vlfuse_t2i_combine", "vlfuse_t2i_combine_kernel"), ("vlfuse_t2i", "vlfuse_t2i_kernel"),
          # (round 5: the FPN output convs run the PLAIN instantiation -- a name of its own, so the DyConv launches are no longer averaged with them)
          ("dcn_igemm8_kernel<16, 0, 1, true", "dcn_igemm8_kernel<plain: FPN convs>"), ("dcn_igemm8_kernelILi16ELi0ELi1ELb1E", "dcn_igemm8_kernel<plain: FPN convs>"),
          # (round 6: the split-precise build runs 8 waves)
          ("dcn_igemm8_kernel<8, 0, 1, true", "dcn_igemm8_kernel<plain: FPN convs>"), ("dcn_igemm8_kernelILi8ELi0ELi1ELb1E", "dcn_igemm8_kernel<plain: FPN convs>"),
          ("dcn_igemm8", "dcn_igemm8_kernel"), ("bert_attn_qkv", "bert_attn_qkv_kernel"), ("gcp_attn_kernel", "gcp_attn_kernel"), ("swin_mlp_kernel<96", "swin_mlp_kernel<96>"), ("swin_mlp_kernel<192", "swin_mlp_kernel<192>"),
          ("swin_mlp_kernel<384", "swin_mlp_kernel<384>"), ("swin_mlp_kernelILi96", "swin_mlp_kernel<96>"), ("swin_mlp_kernelILi192", "swin_mlp_kernel<192>"),
          ("swin_mlp_kernelILi384", "swin_mlp_kernel<384>"), ("swin_mlp2_tail_kernel", "swin_mlp2_tail_kernel"), ("swin_mlp2_kernel<96", "swin_mlp2_kernel<96>"),
          ("swin_mlp2_kernel<192", "swin_mlp2_kernel<192>"), ("swin_mlp2_kernel<384", "swin_mlp2_kernel<384>"), ("dyconv_fuse_group", "dyconv_fuse_group_kernel"), ("dyconv_fuse", "dyconv_fuse_kernel"),
          ("dyrelu_ln", "dyrelu_ln_kernel"), ("layernorm2_kernel", "layernorm2_kernel"), ("layernorm_kernel", "layernorm_kernel"),
          ("window_attn_qkv_kernel<96", "window_attn_qkv_kernel<96>"), ("window_attn_qkv_kernel<192", "window_attn_qkv_kernel<192>"),
          ("window_attn_qkv_kernel<384", "window_attn_qkv_kernel<384>"), ("window_attn_qkv", "window_attn_qkv_kernel"),
          ("attn_resident", "attn_resident_kernel"), ("attn_text", "attn_text_kernel"), ("attn_chunked", "attn_chunked_kernel"),
          ("patch_embed", "patch_embed_kernel"), ("post_select", "post_select_kernel"), ("post_merge", "post_merge_kernel"),
          ("window_attn", "window_attn_kernel"), ("align_fused", "align_fused_kernel"), ("conv3x3_small2", "conv3x3_small2_kernel"), ("conv3x3_group", "conv3x3_group_kernel"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{src}/{counter}.csv")):
        for pat, name in GROUPS:
            if pat in r["Kernel_Name"]:
                agg[name][counter].append(float(r["Counter_Value"]))
                break
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only, eager launches (tools/pmc_traffic.sh)",
       "note": "FETCH_SIZE raw value doubled (gfx950 counts 128-B read requests as 64 B); WRITE_SIZE uncorrected; KB per launch; "
               "traffic_bytes = (2 * fetch_kb_raw + write_kb) * 1024 per launch", "kernels": {}}
for name, d in agg.items():
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    out["kernels"][name] = {"launches_seen": len(d["FETCH_SIZE"]), "fetch_kb_raw": f, "write_kb": w, "traffic_bytes": (2 * f + w) * 1024}
k = out["kernels"]
i2t = k.get("vlfuse_i2t_kernel", {}).get("traffic_bytes", 0.0)
t2i = k.get("vlfuse_t2i_kernel", {}).get("traffic_bytes", 0.0) + k.get("vlfuse_t2i_combine_kernel", {}).get("traffic_bytes", 0.0)
out["vlfuse_traffic_bytes_per_launch_avg"] = int((i2t + t2i) / 2)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
