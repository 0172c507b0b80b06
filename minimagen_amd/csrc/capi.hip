// C-ABI plumbing: error reporting, version, HIP-graph helpers.
#include "common.hip.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void mi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mi_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mi_set_error("%s: %s", what, hipGetErrorString(e));
        return MI_ERR_LAUNCH;
    }
    return MI_OK;
}

extern "C" int mi_abi_version(void) { return MI_ABI_VERSION; }
extern "C" const char* mi_last_error(void) { return g_err; }
extern "C" const char* mi_backend(void) { return MI_BACKEND_STRING; }   // set by the build line
extern "C" int mi_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(mi_act);
        case 1: return (int)sizeof(mi_conv_params);
        case 2: return (int)sizeof(mi_crossembed_params);
        case 3: return (int)sizeof(mi_linear);
        case 4: return (int)sizeof(mi_text_cond_params);
        case 5: return (int)sizeof(mi_cond_step_params);
        case 6: return (int)sizeof(mi_attn_fold_params);
        case 7: return (int)sizeof(mi_cross_attn_params);
        case 8: return (int)sizeof(mi_cfg_x0_params);
        case 9: return (int)sizeof(mi_quantile_params);
        case 10: return (int)sizeof(mi_posterior_params);
        case 11: return (int)sizeof(mi_resize_params);
        case 12: return (int)sizeof(mi_self_attn_params);
        case 13: return (int)sizeof(mi_chan_ff_params);
        case 14: return (int)sizeof(mi_flash_attn_params);
        case 15: return (int)sizeof(mi_tokens_to_nchw_params);
        case 16: return (int)sizeof(mi_conv_wgrad_params);
        case 17: return (int)sizeof(mi_block_bwd_params);
        case 18: return (int)sizeof(mi_crossembed_wgrad_params);
        case 19: return (int)sizeof(mi_folded_attn_params);
        case 20: return (int)sizeof(mi_adam_tensor);
        case 21: return (int)sizeof(mi_adam_params);
        case 22: return (int)sizeof(mi_pack_conv3_desc);
    }
    return -1;
}
