// Library-level glue of libtacotron2_amd.so: ABI version, last-error text, struct sizes.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void t2amd_set_error_(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* t2amd_last_error(void) { return g_err; }
static int g_validate_only = 0;
extern "C" int t2amd_validate_only_flag_(void) { return g_validate_only; }
extern "C" int t2amd_set_validate_only(int on) { g_validate_only = on ? 1 : 0; return T2AMD_OK; }
extern "C" int t2amd_abi_version(void) { return T2AMD_ABI_VERSION; }

extern "C" int t2amd_struct_sizes(int* out, int max_n) {
    const int sizes[] = {
        (int)sizeof(t2amd_gemm_desc), (int)sizeof(t2amd_seg),       (int)sizeof(t2amd_lstm_step),
        (int)sizeof(t2amd_skinny_gemm), (int)sizeof(t2amd_addend),  (int)sizeof(t2amd_lstm_bwd),
        (int)sizeof(t2amd_attn_fwd),  (int)sizeof(t2amd_attn_bwd),  (int)sizeof(t2amd_dec_train),
        (int)sizeof(t2amd_dec_train_bwd), (int)sizeof(t2amd_lstm_seq), (int)sizeof(t2amd_dec_infer),
    };
    const int n = (int)(sizeof(sizes) / sizeof(sizes[0]));
    for (int i = 0; i < n && i < max_n; ++i) out[i] = sizes[i];
    return n;
}
