// vo_mex_common.h -- shared by the MEX gateways: one vo_ctx per MATLAB process, error conversion.
// NOT compiled in the build image (no MATLAB / mex.h there); see INTEGRATION.md for the build line.
#pragma once
#include "mex.h"
#include "../include/vo_hip.h"

static vo_ctx* g_ctx = nullptr;

static void vo_mex_cleanup(void) {
    if (g_ctx) { vo_ctx_destroy(g_ctx); g_ctx = nullptr; }
}

// MATLAB calls MEX on its main thread: one lazily created context, released at exit.
static vo_ctx* vo_mex_ctx(void) {
    if (!g_ctx) {
        int rc = vo_ctx_create(0, &g_ctx);
        if (rc != VO_OK) mexErrMsgIdAndTxt("vo:noDevice", "vo_ctx_create failed: %s", vo_strerror(rc));
        mexLock();
        mexAtExit(vo_mex_cleanup);
    }
    return g_ctx;
}

// mexErrMsgIdAndTxt does not unwind C++ destructors: gateways hold no RAII device temporaries.
static void vo_mex_check(int rc, const char* what) {
    if (rc != VO_OK)
        mexErrMsgIdAndTxt("vo:hip", "%s failed: %s [%s]", what, vo_strerror(rc), vo_ctx_last_error(g_ctx));
}

static const vo_layout VO_MATLAB = {1, 1}; // column-major memory, 1-based pixel coordinates

static double vo_opt(const mxArray* s, const char* name, double dflt) {
    const mxArray* f = s ? mxGetField(s, 0, name) : nullptr;
    return f ? mxGetScalar(f) : dflt;
}
