/* Host-side robustness driver for the C-ABI ("-fsanitize=address on the host C-ABI shim").
 * Built by `make -C diffusion-by-maxentirl_amd/csrc asan` against a HOST-ONLY AddressSanitizer + UBSan build of the
 * library's sources (no device code: hipcc --offload-host-only; never run on the GPU box) and executed by
 * tests/test_cabi_symbols.py.  Every call below hands an entry point a malformed argument set (null pointers, channel
 * counts the kernels cannot tile, negative sizes, over-long lists): the contract (include/dxmi_hip.h, "Conventions") is a
 * negative status + dxmi_last_error() text, no crash, no sanitizer report.  Device pointers are fake non-null addresses: the
 * host side must never dereference them.
 * Output: one line per case "ok <name> <status>" or "FAIL <name> <status>", exit code = number of failures. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "dxmi_hip.h"

static int failures = 0;
#define FAKE(n) ((void*)(uintptr_t)(0x100000ull * (n)))

static void expect_einval(const char* name, int status) {
    const char* msg = dxmi_last_error();
    if (status == DXMI_EINVAL && msg && msg[0]) printf("ok   %-44s %d  (%s)\n", name, status, msg);
    else { printf("FAIL %-44s %d  (%s)\n", name, status, msg ? msg : "(null)"); ++failures; }
}
static void expect_negative(const char* name, int status) {   /* valid arguments, no device: launch / device error, still no crash */
    if (status < 0) printf("ok   %-44s %d\n", name, status);
    else { printf("FAIL %-44s %d\n", name, status); ++failures; }
}

static dxmi_conv_desc good_conv(void) {
    dxmi_conv_desc d;
    memset(&d, 0, sizeof d);
    d.in0 = FAKE(1); d.wpacked = FAKE(2); d.out = FAKE(3);
    d.N = 4; d.IH = d.IW = d.OH = d.OW = 32; d.C0 = 128; d.Cout = 128; d.ksize = 3; d.stride = 1; d.pad = 1;
    return d;
}

int main(void) {
    setvbuf(stdout, NULL, _IOLBF, 0);
    dxmi_conv_desc d;
    /* ---- conv descriptor --------------------------------------------------------------------------------------- */
    expect_einval("conv2d_fwd(NULL desc)", dxmi_conv2d_fwd(NULL, NULL));
    d = good_conv(); d.in0 = NULL;      expect_einval("conv2d_fwd(in0 NULL)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.wpacked = NULL;  expect_einval("conv2d_fwd(wpacked NULL)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.out = NULL;      expect_einval("conv2d_fwd(out NULL)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.N = -4;          expect_einval("conv2d_fwd(N < 0)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.N = 0;           expect_einval("conv2d_fwd(N = 0)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.OH = -32;        expect_einval("conv2d_fwd(OH < 0)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.C0 = 0;          expect_einval("conv2d_fwd(C0 = 0)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.C0 = 100;        expect_einval("conv2d_fwd(C0 % 32 != 0)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.Cout = -128;     expect_einval("conv2d_fwd(Cout < 0)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.ksize = 5;       expect_einval("conv2d_fwd(ksize 5)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.stride = 3;      expect_einval("conv2d_fwd(stride 3)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.C1 = 128;        expect_einval("conv2d_fwd(C1 > 0, in1 NULL)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.act = 17;        expect_einval("conv2d_fwd(act 17)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.addvec = (const float*)FAKE(4); d.addvec_ld = 64;
                                        expect_einval("conv2d_fwd(addvec_ld < Cout)", dxmi_conv2d_fwd(&d, NULL));
    /* the wave-specialised path needs Cout % 64 == 0: variant pins it, so the dispatcher cannot quietly re-route */
    d = good_conv(); d.Cout = 96; d.gn_stats = (float*)FAKE(5);
                                        expect_einval("conv2d_fwd(gn_stats, Cout % 64 != 0)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.gn_out = FAKE(6); d.gn_groups = 32;     /* 32x32 map: no kernel holds whole images */
                                        expect_einval("conv2d_fwd(gn_out on a 32x32 map)", dxmi_conv2d_fwd(&d, NULL));
    d = good_conv(); d.IH = d.IW = d.OH = d.OW = 4; d.C0 = d.Cout = 256; d.gn_out = FAKE(6); d.gn_groups = 32;
                                        expect_einval("conv2d_fwd(gn_out, gamma NULL)", dxmi_conv2d_fwd(&d, NULL));
    expect_einval("conv2d_kernel_id(NULL)", dxmi_conv2d_kernel_id(NULL) < 0 ? DXMI_EINVAL : 0);
    if (dxmi_conv2d_gn_fuse_supported(NULL) != 0 || dxmi_conv2d_gn_stats_partials(NULL) != 0) { printf("FAIL gn queries(NULL)\n"); ++failures; }
    else printf("ok   gn_fuse_supported / gn_stats_partials(NULL) -> 0\n");
    d = good_conv();                    expect_negative("conv2d_fwd(valid desc, no device)", dxmi_conv2d_fwd(&d, NULL));

    /* ---- weight packing ---------------------------------------------------------------------------------------- */
    expect_einval("pack_conv_weight(NULL src)", dxmi_pack_conv_weight(NULL, FAKE(1), 128, 128, 3, 0, 0, NULL));
    expect_einval("pack_conv_weight(NULL dst)", dxmi_pack_conv_weight((const float*)FAKE(1), NULL, 128, 128, 3, 0, 0, NULL));
    expect_einval("pack_conv_weight(ksize 2)", dxmi_pack_conv_weight((const float*)FAKE(1), FAKE(2), 128, 128, 2, 0, 0, NULL));
    expect_einval("pack_conv_weight(Cout < 0)", dxmi_pack_conv_weight((const float*)FAKE(1), FAKE(2), -128, 128, 3, 0, 0, NULL));
    expect_einval("pack_conv_weights(NULL items)", dxmi_pack_conv_weights(NULL, 3, NULL));
    { dxmi_pack_item it[2]; memset(it, 0, sizeof it);
      it[0].w = (const float*)FAKE(1); it[0].dst = FAKE(2); it[0].Cout = it[0].Cin = 128; it[0].ksize = 3;
      it[1] = it[0]; it[1].dst = NULL;
      expect_einval("pack_conv_weights(item dst NULL)", dxmi_pack_conv_weights(it, 2, NULL));
      expect_einval("pack_conv_weights(count < 0)", dxmi_pack_conv_weights(it, -1, NULL)); }
    if (dxmi_packed_conv_weight_bytes(-1, 128, 3, 0) > 0 || dxmi_packed_conv_weight_bytes(128, 128, 7, 0) > 0) { printf("FAIL packed_conv_weight_bytes(bad)\n"); ++failures; }
    else printf("ok   packed_conv_weight_bytes(bad) <= 0\n");

    /* ---- weight gradient --------------------------------------------------------------------------------------- */
    expect_einval("conv2d_wgrad(NULL x)", dxmi_conv2d_wgrad(NULL, 128, NULL, 0, FAKE(1), (float*)FAKE(2), FAKE(3), 4, 32, 32, 32, 32, 128, 3, 1, 1, 0, 0, NULL));
    expect_einval("conv2d_wgrad(NULL workspace)", dxmi_conv2d_wgrad(FAKE(1), 128, NULL, 0, FAKE(1), (float*)FAKE(2), NULL, 4, 32, 32, 32, 32, 128, 3, 1, 1, 0, 0, NULL));
    expect_einval("conv2d_wgrad(Cout % 64 != 0)", dxmi_conv2d_wgrad(FAKE(1), 128, NULL, 0, FAKE(1), (float*)FAKE(2), FAKE(3), 4, 32, 32, 32, 32, 96, 3, 1, 1, 0, 0, NULL));
    expect_einval("conv2d_wgrad(C0 = 0)", dxmi_conv2d_wgrad(FAKE(1), 0, NULL, 0, FAKE(1), (float*)FAKE(2), FAKE(3), 4, 32, 32, 32, 32, 128, 3, 1, 1, 0, 0, NULL));
    expect_einval("conv2d_wgrad(N < 0)", dxmi_conv2d_wgrad(FAKE(1), 128, NULL, 0, FAKE(1), (float*)FAKE(2), FAKE(3), -4, 32, 32, 32, 32, 128, 3, 1, 1, 0, 0, NULL));
    expect_einval("conv2d_wgrad(OW = 24)", dxmi_conv2d_wgrad(FAKE(1), 128, NULL, 0, FAKE(1), (float*)FAKE(2), FAKE(3), 4, 24, 24, 24, 24, 128, 3, 1, 1, 0, 0, NULL));
    expect_einval("conv2d_wgrad(OH < 0)", dxmi_conv2d_wgrad(FAKE(1), 128, NULL, 0, FAKE(1), (float*)FAKE(2), FAKE(3), 4, 32, 32, -32, 32, 128, 3, 1, 1, 0, 0, NULL));
    expect_einval("conv2d_wgrad_bias(NULL dbias)", dxmi_conv2d_wgrad_bias(FAKE(1), 128, NULL, 0, FAKE(1), (float*)FAKE(2), NULL, FAKE(3), 4, 32, 32, 32, 32, 128, 3, 1, 1, 0, 0, NULL));
    if (dxmi_conv2d_wgrad_workspace_bytes(16, 16, 16, 256, 256, 1) < 64ll * 256 * 256 * 4 + 64ll * 4 * 256 * 4) { printf("FAIL wgrad_workspace_bytes(1x1, 64 splits)\n"); ++failures; }
    else printf("ok   wgrad_workspace_bytes covers the 64 splits of the 128 x 128 1x1 kernel\n");

    /* ---- GroupNorm / attention / elementwise / optimiser lists ------------------------------------------------- */
    expect_einval("groupnorm_silu_fwd(NULL in)", dxmi_groupnorm_silu_fwd(NULL, 128, NULL, 0, (const float*)FAKE(1), (const float*)FAKE(2), FAKE(3), 4, 1024, 32, 1e-6f, 1, NULL));
    expect_einval("groupnorm_silu_fwd(groups = 0)", dxmi_groupnorm_silu_fwd(FAKE(1), 128, NULL, 0, (const float*)FAKE(1), (const float*)FAKE(2), FAKE(3), 4, 1024, 0, 1e-6f, 1, NULL));
    expect_einval("groupnorm_silu_fwd(C % groups)", dxmi_groupnorm_silu_fwd(FAKE(1), 100, NULL, 0, (const float*)FAKE(1), (const float*)FAKE(2), FAKE(3), 4, 1024, 32, 1e-6f, 1, NULL));
    expect_einval("groupnorm_apply(NULL stats)", dxmi_groupnorm_apply(FAKE(1), 128, NULL, 8, NULL, 0, NULL, 0, (const float*)FAKE(1), (const float*)FAKE(2), NULL, 0, FAKE(3), 4, 1024, 32, 1e-6f, 1, NULL));
    expect_einval("groupnorm_generic_fwd(N < 0)", dxmi_groupnorm_generic_fwd(FAKE(1), 192, NULL, 0, (const float*)FAKE(1), (const float*)FAKE(2), NULL, 0, FAKE(3), FAKE(4), -1, 1024, 32, 1e-5f, 1, NULL));
    expect_einval("attention_fwd(NULL qkv)", dxmi_attention_fwd(NULL, FAKE(1), 4, 256, 256, 1, 0.0625f, NULL));
    expect_einval("attention_fwd(C % heads)", dxmi_attention_fwd(FAKE(1), FAKE(2), 4, 256, 250, 4, 0.0625f, NULL));
    expect_einval("attention_fwd(heads = 0)", dxmi_attention_fwd(FAKE(1), FAKE(2), 4, 256, 256, 0, 0.0625f, NULL));
    expect_einval("attention_proj_fwd(T = 64)", dxmi_attention_proj_fwd(FAKE(1), FAKE(2), (const float*)FAKE(3), FAKE(4), FAKE(5), NULL, 4, 64, 256, 1, 0.0625f, NULL));
    /* round 5: the AttnBlock as one launch */
    expect_einval("attn_block_fwd(x NULL)", dxmi_attn_block_fwd(NULL, (const float*)FAKE(2), 2, (const float*)FAKE(3), (const float*)FAKE(4), 1e-6f, FAKE(5), FAKE(6), NULL, 4, 256, 256, NULL));
    expect_einval("attn_block_fwd(stats NULL)", dxmi_attn_block_fwd(FAKE(1), NULL, 2, (const float*)FAKE(3), (const float*)FAKE(4), 1e-6f, FAKE(5), FAKE(6), NULL, 4, 256, 256, NULL));
    expect_einval("attn_block_fwd(T = 64)", dxmi_attn_block_fwd(FAKE(1), (const float*)FAKE(2), 2, (const float*)FAKE(3), (const float*)FAKE(4), 1e-6f, FAKE(5), FAKE(6), NULL, 4, 64, 256, NULL));
    expect_einval("attn_block_fwd(C = 512)", dxmi_attn_block_fwd(FAKE(1), (const float*)FAKE(2), 2, (const float*)FAKE(3), (const float*)FAKE(4), 1e-6f, FAKE(5), FAKE(6), NULL, 4, 256, 512, NULL));
    expect_einval("attn_block_fwd(P = 9)", dxmi_attn_block_fwd(FAKE(1), (const float*)FAKE(2), 9, (const float*)FAKE(3), (const float*)FAKE(4), 1e-6f, FAKE(5), FAKE(6), NULL, 4, 256, 256, NULL));
    expect_einval("attn_block_fwd(P = 0)", dxmi_attn_block_fwd(FAKE(1), (const float*)FAKE(2), 0, (const float*)FAKE(3), (const float*)FAKE(4), 1e-6f, FAKE(5), FAKE(6), NULL, 4, 256, 256, NULL));
    expect_einval("attn_block_fwd(N = 0)", dxmi_attn_block_fwd(FAKE(1), (const float*)FAKE(2), 2, (const float*)FAKE(3), (const float*)FAKE(4), 1e-6f, FAKE(5), FAKE(6), NULL, 0, 256, 256, NULL));
    expect_einval("attn_block_pack(wq NULL)", dxmi_attn_block_pack(NULL, (const float*)FAKE(2), (const float*)FAKE(3), (const float*)FAKE(4), (const float*)FAKE(5), (const float*)FAKE(6), (const float*)FAKE(7), 0.0625f, FAKE(8), NULL));
    expect_einval("attn_block_pack(dst NULL)", dxmi_attn_block_pack((const float*)FAKE(1), (const float*)FAKE(2), (const float*)FAKE(3), (const float*)FAKE(4), (const float*)FAKE(5), (const float*)FAKE(6), (const float*)FAKE(7), 0.0625f, NULL, NULL));
    expect_negative("attn_block_fwd(valid, no device)", dxmi_attn_block_fwd(FAKE(1), (const float*)FAKE(2), 2, (const float*)FAKE(3), (const float*)FAKE(4), 1e-6f, FAKE(5), FAKE(6), NULL, 4, 256, 256, NULL));
    expect_einval("var_step_fwd(NULL x)", dxmi_var_step_fwd(NULL, (const float*)FAKE(1), (const float*)FAKE(2), (const float*)FAKE(3), (const float*)FAKE(4), (const float*)FAKE(5), (float*)FAKE(6), (float*)FAKE(7), (float*)FAKE(8), (float*)FAKE(9), 4, 3072, 0, NULL));
    expect_einval("var_step_fwd(N < 0)", dxmi_var_step_fwd((const float*)FAKE(1), (const float*)FAKE(1), (const float*)FAKE(2), (const float*)FAKE(3), (const float*)FAKE(4), (const float*)FAKE(5), (float*)FAKE(6), (float*)FAKE(7), (float*)FAKE(8), (float*)FAKE(9), -4, 3072, 0, NULL));
    expect_einval("var_gather_sched(NULL t)", dxmi_var_gather_sched(NULL, (const float*)FAKE(1), (const float*)FAKE(2), (const float*)FAKE(3), (const float*)FAKE(4), (float*)FAKE(5), (float*)FAKE(6), (float*)FAKE(7), (float*)FAKE(8), 4, 10, NULL));
    expect_einval("gather_rows(NULL idx)", dxmi_gather_rows(FAKE(1), NULL, FAKE(2), 16, 64, 12288, NULL));
    expect_einval("gather_rows(row_bytes < 0)", dxmi_gather_rows(FAKE(1), (const int64_t*)FAKE(3), FAKE(2), 16, 64, -12288, NULL));
    expect_einval("quantize_u8(mode 9)", dxmi_quantize_u8((const float*)FAKE(1), FAKE(2), 4, 3, 1024, 9, 1, NULL));
    expect_einval("linear_fwd(K = 0)", dxmi_linear_fwd((const float*)FAKE(1), FAKE(2), NULL, (float*)FAKE(3), 4, 0, 512, 0, 0, NULL));
    expect_einval("timestep_embedding(dim odd)", dxmi_timestep_embedding((const float*)FAKE(1), (float*)FAKE(2), 4, 127, 0, 10000.f, NULL));
    { void* ptrs[2] = {FAKE(1), FAKE(2)}; int64_t numel[2] = {1024, -5}; float lr[2] = {1e-3f, 1e-3f};
      expect_einval("adam_step(count > DXMI_MT_MAX)", dxmi_adam_step(ptrs, ptrs, ptrs, ptrs, numel, lr, DXMI_MT_MAX + 1, 0.9, 0.999, 1e-8, 1.0, NULL, 0, NULL));
      expect_einval("adam_step(NULL lists)", dxmi_adam_step(NULL, ptrs, ptrs, ptrs, numel, lr, 2, 0.9, 0.999, 1e-8, 1.0, NULL, 0, NULL));
      expect_einval("adam_step(numel < 0)", dxmi_adam_step(ptrs, ptrs, ptrs, ptrs, numel, lr, 2, 0.9, 0.999, 1e-8, 1.0, NULL, 0, NULL));
      expect_einval("radam_step(count < 0)", dxmi_radam_step(ptrs, ptrs, ptrs, ptrs, numel, lr, -2, 0.9, 0.999, 1e-8, 0.1, 1.0, 0.0, NULL, NULL, NULL));
      expect_einval("gradnorm_clip(NULL partials)", dxmi_gradnorm_clip(ptrs, numel, 1, 0.1f, NULL, (float*)FAKE(3), 1, NULL)); }
    expect_einval("dropout_bf16(p = 1.5)", dxmi_dropout_bf16(FAKE(1), FAKE(2), 1024, 1.5f, 7u, NULL));
    expect_einval("colsum_f32(B = 0)", dxmi_colsum_f32((const float*)FAKE(1), (float*)FAKE(2), 0, 4, 128, NULL));
    expect_einval("edm_step_fwd(NULL sigma)", dxmi_edm_step_fwd((const float*)FAKE(1), (const float*)FAKE(2), (const float*)FAKE(3), NULL, (const float*)FAKE(4), (const float*)FAKE(5), (float*)FAKE(6), (float*)FAKE(7), 4, 12288, 0.5f, NULL));

    expect_einval("attention_bwd(NULL o)", dxmi_attention_bwd(FAKE(1), NULL, FAKE(2), FAKE(3), FAKE(4), 2, 256, 128, 2, 0.125f, NULL));
    expect_einval("attention_bwd(head dim 32)", dxmi_attention_bwd(FAKE(1), FAKE(5), FAKE(2), FAKE(3), FAKE(4), 2, 256, 128, 4, 0.125f, NULL));
    expect_einval("attention_bwd(T = 0)", dxmi_attention_bwd(FAKE(1), FAKE(5), FAKE(2), FAKE(3), FAKE(4), 2, 0, 128, 2, 0.125f, NULL));
    if (dxmi_attention_bwd_supported(256, 256, 1) != 0 || dxmi_attention_bwd_supported(1024, 384, 6) != 1 || dxmi_attention_bwd_workspace_bytes(0, 256, 2) != 0) { printf("FAIL attention_bwd queries\n"); ++failures; }
    else printf("ok   attention_bwd_supported / workspace_bytes\n");

    /* ---- FID statistics ------------------------------------------------------------------------------------------ */
    expect_einval("fid_stats(NULL act)", dxmi_fid_stats(NULL, 100, 64, (double*)FAKE(1), (double*)FAKE(2), FAKE(3), NULL));
    expect_einval("fid_stats(N = 1)", dxmi_fid_stats((const float*)FAKE(1), 1, 64, (double*)FAKE(1), (double*)FAKE(2), FAKE(3), NULL));
    expect_einval("fid_stats(D % 4 != 0)", dxmi_fid_stats((const float*)FAKE(1), 100, 66, (double*)FAKE(1), (double*)FAKE(2), FAKE(3), NULL));
    expect_einval("fid_stats(D < 0)", dxmi_fid_stats((const float*)FAKE(1), 100, -64, (double*)FAKE(1), (double*)FAKE(2), FAKE(3), NULL));
    expect_einval("fid_stats(NULL workspace)", dxmi_fid_stats((const float*)FAKE(1), 100, 64, (double*)FAKE(1), (double*)FAKE(2), NULL, NULL));
    if (dxmi_fid_stats_workspace_bytes(-5, 64) != 0 || dxmi_fid_stats_workspace_bytes(100, -64) != 0) { printf("FAIL fid_stats_workspace_bytes(bad)\n"); ++failures; }
    else printf("ok   fid_stats_workspace_bytes(bad) == 0\n");

    expect_einval("groupnorm_generic_bwd_saved(NULL dy)", dxmi_groupnorm_generic_bwd_saved(FAKE(1), 192, NULL, 0, NULL, NULL, NULL, (const float*)FAKE(1), (const float*)FAKE(2), NULL, 0, FAKE(3), NULL, (float*)FAKE(4), (const float*)FAKE(5), FAKE(6), 2, 1024, 32, 1e-5f, 1, NULL));
    expect_einval("groupnorm_generic_bwd_saved(HW = 0)", dxmi_groupnorm_generic_bwd_saved(FAKE(1), 192, NULL, 0, FAKE(2), NULL, NULL, (const float*)FAKE(1), (const float*)FAKE(2), NULL, 0, FAKE(3), NULL, (float*)FAKE(4), (const float*)FAKE(5), FAKE(6), 2, 0, 32, 1e-5f, 1, NULL));
    expect_einval("linear_splitk(NULL partials)", dxmi_linear_splitk((const float*)FAKE(1), FAKE(2), NULL, 16, 30720, 768, 0, NULL));
    expect_einval("linear_splitk(M % 4 != 0)", dxmi_linear_splitk((const float*)FAKE(1), FAKE(2), (float*)FAKE(3), 16, 30720, 770, 0, NULL));
    expect_einval("linear_splitk(K % 32 != 0)", dxmi_linear_splitk((const float*)FAKE(1), FAKE(2), (float*)FAKE(3), 16, 30700, 768, 0, NULL));
    if (dxmi_linear_splitk_slices(-1, 4096, 64) != 0 || dxmi_linear_splitk_slices(16, 0, 64) != 0 || dxmi_linear_splitk_slices(16, 30720, 768) < 2) { printf("FAIL linear_splitk_slices\n"); ++failures; }
    else printf("ok   linear_splitk_slices(bad) == 0, (16, 30720, 768) splits\n");
    /* ---- round 6: replayed-step entry points, fused TD step, InceptionV3 ops ---------------------------------------- */
    {
        void* ptrs[2] = {FAKE(1), FAKE(2)};
        int64_t numel[2] = {1024, 4096};
        int64_t bad_numel[2] = {1024, 0};
        expect_einval("adam_step_dev(NULL hyper)", dxmi_adam_step_dev(ptrs, ptrs, ptrs, ptrs, numel, 2, 0.9, 0.999, 1e-8, NULL, NULL, 0, NULL));
        expect_einval("adam_step_dev(count = 0)", dxmi_adam_step_dev(ptrs, ptrs, ptrs, ptrs, numel, 0, 0.9, 0.999, 1e-8, (const float*)FAKE(3), NULL, 0, NULL));
        expect_einval("adam_step_dev(empty tensor)", dxmi_adam_step_dev(ptrs, ptrs, ptrs, ptrs, bad_numel, 2, 0.9, 0.999, 1e-8, (const float*)FAKE(3), NULL, 0, NULL));
        expect_einval("radam_step_dev(NULL grads)", dxmi_radam_step_dev(ptrs, NULL, ptrs, ptrs, numel, 2, 0.9, 0.999, 1e-8, (const float*)FAKE(3), NULL, NULL, NULL));
        expect_einval("dropout_bf16_dev(NULL seed)", dxmi_dropout_bf16_dev(FAKE(1), FAKE(2), 1024, 0.1f, NULL, NULL));
        expect_einval("dropout_bf16_dev(n % 8 != 0)", dxmi_dropout_bf16_dev(FAKE(1), FAKE(2), 1020, 0.1f, (const uint32_t*)FAKE(3), NULL));
        expect_einval("td_gather_cost(no next source)", dxmi_td_gather_cost((const float*)FAKE(1), (const int64_t*)FAKE(2), NULL, NULL, (const float*)FAKE(3), (float*)FAKE(4), (float*)FAKE(5), (float*)FAKE(6), 32, 3072, 352, NULL));
        expect_einval("td_gather_cost(CHW % 4 != 0)", dxmi_td_gather_cost((const float*)FAKE(1), (const int64_t*)FAKE(2), (const int64_t*)FAKE(7), NULL, (const float*)FAKE(3), (float*)FAKE(4), (float*)FAKE(5), (float*)FAKE(6), 32, 3070, 352, NULL));
        expect_einval("td_gather_cost(NULL beta)", dxmi_td_gather_cost((const float*)FAKE(1), (const int64_t*)FAKE(2), (const int64_t*)FAKE(7), NULL, NULL, (float*)FAKE(4), (float*)FAKE(5), (float*)FAKE(6), 32, 3072, 352, NULL));
        expect_einval("gn_blockstats_to_generic(NULL stats)", dxmi_gn_blockstats_to_generic(NULL, 4, 192, NULL, 0, 0, (float*)FAKE(2), 4, 1024, 32, NULL));
        expect_einval("gn_blockstats_to_generic(odd channels per group)", dxmi_gn_blockstats_to_generic((const float*)FAKE(1), 4, 96, NULL, 0, 0, (float*)FAKE(2), 4, 1024, 32, NULL));
        expect_einval("gn_blockstats_to_generic(second part without stats)", dxmi_gn_blockstats_to_generic((const float*)FAKE(1), 4, 192, NULL, 0, 192, (float*)FAKE(2), 4, 1024, 32, NULL));
        expect_einval("gn_ss_grads(NULL g)", dxmi_gn_ss_grads(NULL, (const float*)FAKE(1), 768, (const float*)FAKE(2), (const float*)FAKE(3), (float*)FAKE(4), (float*)FAKE(5), (float*)FAKE(6), 16, 384, NULL));
        expect_einval("gn_ss_grads(ss_ld < 2C)", dxmi_gn_ss_grads((const float*)FAKE(7), (const float*)FAKE(1), 384, (const float*)FAKE(2), (const float*)FAKE(3), (float*)FAKE(4), (float*)FAKE(5), (float*)FAKE(6), 16, 384, NULL));
        expect_einval("attention_fwd_lse(NULL lse)", dxmi_attention_fwd_lse(FAKE(1), FAKE(2), NULL, 2, 256, 384, 6, 0.125f, NULL));
        expect_einval("attention_fwd_lse(256 x 256 single head)", dxmi_attention_fwd_lse(FAKE(1), FAKE(2), (float*)FAKE(3), 2, 256, 256, 1, 0.0625f, NULL));
        expect_einval("attention_bwd_lse(NULL lse)", dxmi_attention_bwd_lse(FAKE(1), FAKE(2), FAKE(3), FAKE(4), NULL, FAKE(5), 2, 256, 384, 6, 0.125f, NULL));
        expect_einval("td_loss(B = 0)", dxmi_td_loss((const float*)FAKE(1), (const float*)FAKE(2), NULL, (float*)FAKE(3), (float*)FAKE(4), 0, NULL));
        expect_einval("value_head_pgrad(NULL dy)", dxmi_value_head_pgrad((const float*)FAKE(1), (const float*)FAKE(2), (const float*)FAKE(3), NULL, NULL, (float*)FAKE(4), 32, 256, NULL));
        expect_einval("gconv_fwd(Cin % 16 != 0)", dxmi_gconv_fwd(FAKE(1), FAKE(2), (const float*)FAKE(3), FAKE(4), 2, 35, 35, 40, 64, 5, 5, 1, 1, 2, 2, 64, 0, 1, NULL));
        expect_einval("gconv_fwd(kernel larger than the padded map)", dxmi_gconv_fwd(FAKE(1), FAKE(2), (const float*)FAKE(3), FAKE(4), 2, 3, 3, 16, 32, 7, 7, 1, 1, 0, 0, 32, 0, 1, NULL));
        expect_einval("gconv_fwd(channel window outside the output)", dxmi_gconv_fwd(FAKE(1), FAKE(2), (const float*)FAKE(3), FAKE(4), 2, 17, 17, 768, 192, 1, 1, 1, 1, 0, 0, 768, 600, 1, NULL));
        expect_einval("gconv_fwd(3x3 s2 on a 2x2 map)", dxmi_gconv_fwd(FAKE(1), FAKE(2), (const float*)FAKE(3), FAKE(4), 2, 2, 2, 16, 32, 3, 3, 2, 2, 0, 0, 32, 0, 1, NULL));
        expect_einval("gconv_fwd(stride 0)", dxmi_gconv_fwd(FAKE(1), FAKE(2), (const float*)FAKE(3), FAKE(4), 2, 17, 17, 768, 192, 1, 1, 0, 1, 0, 0, 192, 0, 1, NULL));
        expect_einval("gconv_pack(gamma without var)", dxmi_gconv_pack((const float*)FAKE(1), (const float*)FAKE(2), (const float*)FAKE(3), (const float*)FAKE(4), NULL, 1e-3f, FAKE(5), (float*)FAKE(6), 64, 48, 5, 5, NULL));
        expect_einval("pool3x3(C % 8 != 0)", dxmi_pool3x3(FAKE(1), FAKE(2), 2, 35, 35, 100, 1, 1, 1, 104, 0, NULL));
        expect_einval("pool3x3(2x2 map, no padding)", dxmi_pool3x3(FAKE(1), FAKE(2), 2, 2, 2, 64, 2, 0, 0, 64, 0, NULL));
        expect_einval("global_avgpool(HW = 0)", dxmi_global_avgpool(FAKE(1), (float*)FAKE(2), 2, 0, 2048, NULL));
        expect_einval("resize_bilinear_nhwc16(OH < 0)", dxmi_resize_bilinear_nhwc16((const float*)FAKE(1), FAKE(2), 2, 32, 32, -299, 299, 1, NULL));
        if (dxmi_gconv_packed_elems(80, 64, 1, 1) != 96 * 64 || dxmi_gconv_packed_elems(32, 3, 3, 3) != 32 * 9 * 16) { printf("FAIL gconv_packed_elems\n"); ++failures; }
        else printf("ok   gconv_packed_elems pads couts to 32 and channels to 16\n");
    }
    /* ---- kernel-selection knobs ---------------------------------------------------------------------------------- */
    {
        int32_t v = -1;
        expect_einval("set_tuning(NULL name)", dxmi_set_tuning(NULL, 1));
        expect_einval("set_tuning(unknown knob)", dxmi_set_tuning("no_such_knob", 1));
        expect_einval("set_tuning(negative value)", dxmi_set_tuning("conv_ws_min_tiles", -3));
        expect_einval("get_tuning(NULL result)", dxmi_get_tuning("conv_ws_min_tiles", NULL));
        expect_einval("get_tuning(unknown knob)", dxmi_get_tuning("", &v));
        if (dxmi_set_tuning("conv_ws_min_tiles", 7) != 0 || dxmi_get_tuning("conv_ws_min_tiles", &v) != 0 || v != 7) { printf("FAIL tuning round trip\n"); ++failures; }
        else printf("ok   set_tuning / get_tuning round trip\n");
        d = good_conv();                /* 16 tiles: the knob decides between the two kernels, neither may touch the fake pointers */
        dxmi_set_tuning("conv_ws_min_tiles", 0);
        expect_negative("conv2d_fwd(valid desc on the wave-specialised kernel, no device)", dxmi_conv2d_fwd(&d, NULL));
        dxmi_set_tuning("conv_ws_min_tiles", 96);
    }

    printf("%d failure(s)\n", failures);
    return failures > 99 ? 99 : failures;
}
