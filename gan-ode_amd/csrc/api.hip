// api.hip -- program runner and ABI introspection of libgode.so
#include "common.h"

extern "C" int gode_version(void) { return GODE_VERSION; }

extern "C" int gode_sizeof(int kind) {
  switch (kind) {
    case 0: return (int)sizeof(gode_conv_geom);
    case GODE_OP_IGEMM: return (int)sizeof(gode_igemm_op);
    case GODE_OP_WGRAD: return (int)sizeof(gode_wgrad_op);
    case GODE_OP_BN_FINALIZE: return (int)sizeof(gode_bn_finalize_op);
    case GODE_OP_BN_BWD: return (int)sizeof(gode_bn_bwd_op);
    case GODE_OP_ODE_FWD: return (int)sizeof(gode_ode_fwd_op);
    case GODE_OP_ODE_BWD: return (int)sizeof(gode_ode_bwd_op);
    case GODE_OP_BCE: return (int)sizeof(gode_bce_op);
    case GODE_OP_ADAM: return (int)sizeof(gode_adam_op);
    case GODE_OP_PACK: return (int)sizeof(gode_pack_op);
    case GODE_OP_ODERNN_FWD: return (int)sizeof(gode_odernn_fwd_op);
    case GODE_OP_ODERNN_BWD: return (int)sizeof(gode_odernn_bwd_op);
    case GODE_OP_BN_APPLY: return (int)sizeof(gode_bn_apply_op);
    case GODE_OP_COL2IM: return (int)sizeof(gode_col2im_op);
  }
  return GODE_E_KIND;
}

// Executes a pre-built list of ops back to back on one stream: one host->library call per network pass keeps the
// launch path short at batch 32, where a whole generator forward is ~1 ms of GPU time.
extern "C" int gode_pack_batch_(const gode_pack_op* const* ops, int n, void* stream);   // igemm.hip

extern "C" int gode_run(const int32_t* kinds, const void* const* ops, int32_t n, void* stream) {
  if (n < 0 || (n > 0 && (!kinds || !ops))) return GODE_E_ARG;
  for (int i = 0; i < n; ++i) {
    int rc;
    switch (kinds[i]) {
      case GODE_OP_IGEMM: rc = gode_igemm((const gode_igemm_op*)ops[i], stream); break;
      case GODE_OP_WGRAD: rc = gode_wgrad((const gode_wgrad_op*)ops[i], stream); break;
      case GODE_OP_BN_FINALIZE: rc = gode_bn_finalize((const gode_bn_finalize_op*)ops[i], stream); break;
      case GODE_OP_BN_BWD: rc = gode_bn_bwd((const gode_bn_bwd_op*)ops[i], stream); break;
      case GODE_OP_ODE_FWD: rc = gode_ode_fwd((const gode_ode_fwd_op*)ops[i], stream); break;
      case GODE_OP_ODE_BWD: rc = gode_ode_bwd((const gode_ode_bwd_op*)ops[i], stream); break;
      case GODE_OP_BCE: rc = gode_bce_logits((const gode_bce_op*)ops[i], stream); break;
      case GODE_OP_ADAM: rc = gode_adam_l2((const gode_adam_op*)ops[i], stream); break;
      case GODE_OP_ODERNN_FWD: rc = gode_odernn_fwd((const gode_odernn_fwd_op*)ops[i], stream); break;
      case GODE_OP_ODERNN_BWD: rc = gode_odernn_bwd((const gode_odernn_bwd_op*)ops[i], stream); break;
      case GODE_OP_BN_APPLY: rc = gode_bn_apply((const gode_bn_apply_op*)ops[i], stream); break;
      case GODE_OP_COL2IM: rc = gode_col2im((const gode_col2im_op*)ops[i], stream); break;
      case GODE_OP_PACK: {
        // every run of consecutive pack ops goes out as one launch per 8 panels
        int j = i;
        while (j + 1 < n && kinds[j + 1] == GODE_OP_PACK) ++j;
        rc = gode_pack_batch_((const gode_pack_op* const*)(ops + i), j - i + 1, stream);
        if (rc == 0) i = j;
        break;
      }
      default: return GODE_E_KIND;
    }
    if (rc != 0) return rc > 0 ? rc : rc * 1000 - i;  // negative codes carry the op index: -(code*1000 + i)
  }
  return 0;
}
