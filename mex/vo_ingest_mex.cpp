// The decode pool (vo_ingest_*) for MATLAB: imread / rgb2gray of a whole sequence off the interpreter's thread
// (main.m:15-30,52-59; utils/dataLoaderKitti.m:29).
//   ing = vo_ingest_mex('open', files, opts)      files: cell array of char; opts: threads, depth, keep_rgb, hold
//                                                 (hold = 2: the frame before the current one stays valid too)
//   [ok, dimg, H, W] = vo_ingest_mex('next', ing) dimg: uint64 device handle of an H-rows-of-W-bytes gray image,
//                                                 valid until the next call (the one after it with hold = 2);
//                                                 ok = false after the last file
//   img = vo_ingest_mex('read', ing)              the current frame as a MATLAB H x W uint8 array (tests, plots)
//   vo_ingest_mex('close', ing)
#include <string.h>

#include <string>
#include <vector>

#include "vo_mex_common.h"

static vo_ingest* pool_of(const mxArray* a) { return (vo_ingest*)(uintptr_t)(*(uint64_t*)mxGetData(a)); }
static int g_h = 0, g_w = 0, g_ch = 0;

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    char cmd[32];
    if (nrhs < 2 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("vo:args", "usage: vo_ingest_mex(command, ...)");
    vo_ctx* ctx = vo_mex_ctx();
    if (!strcmp(cmd, "open")) {
        const mxArray* files = prhs[1];
        const int n = (int)mxGetNumberOfElements(files);
        std::vector<std::string> names(n);
        std::vector<const char*> ptrs(n);
        for (int i = 0; i < n; ++i) {
            char* s = mxArrayToString(mxGetCell(files, i));
            names[i] = s ? s : "";
            mxFree(s);
            ptrs[i] = names[i].c_str();
        }
        vo_ingest_opts o = {0, 0, 0};
        if (nrhs > 2) { o.threads = (int)vo_opt(prhs[2], "threads", 0); o.depth = (int)vo_opt(prhs[2], "depth", 0); o.keep_rgb = (int)vo_opt(prhs[2], "keep_rgb", 0); }
        vo_ingest* g = nullptr;
        vo_mex_check(vo_ingest_open(ctx, ptrs.data(), n, &o, &g), "vo_ingest_open");
        const int hold = nrhs > 2 ? (int)vo_opt(prhs[2], "hold", 1) : 1;
        if (hold != 1) {
            const int rc = vo_ingest_set_hold(g, hold);
            if (rc != VO_OK) { (void)vo_ingest_close(g); vo_mex_check(rc, "vo_ingest_set_hold"); }
        }
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *(uint64_t*)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)g;
    } else if (!strcmp(cmd, "next")) {
        const uint8_t* d = nullptr;
        int idx = 0;
        const int rc = vo_ingest_next(pool_of(prhs[1]), &d, &g_h, &g_w, &g_ch, &idx);
        if (rc != VO_OK && rc != VO_END) vo_mex_check(rc, "vo_ingest_next");
        plhs[0] = mxCreateLogicalScalar(rc == VO_OK);
        if (nlhs > 1) { plhs[1] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL); *(uint64_t*)mxGetData(plhs[1]) = (uint64_t)(uintptr_t)d; }
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(g_h);
        if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(g_w);
    } else if (!strcmp(cmd, "read")) {
        std::vector<uint8_t> rows((size_t)g_h * g_w * g_ch);
        vo_mex_check(vo_ingest_read(pool_of(prhs[1]), rows.data(), rows.size()), "vo_ingest_read");
        const mwSize dims[3] = {(mwSize)g_h, (mwSize)g_w, (mwSize)g_ch};
        plhs[0] = mxCreateNumericArray(g_ch == 1 ? 2 : 3, dims, mxUINT8_CLASS, mxREAL);
        uint8_t* o = (uint8_t*)mxGetData(plhs[0]);   // interleaved rows -> MATLAB's column-major planes
        for (int c = 0; c < g_ch; ++c)
            for (int x = 0; x < g_w; ++x)
                for (int y = 0; y < g_h; ++y) o[((size_t)c * g_w + x) * g_h + y] = rows[((size_t)y * g_w + x) * g_ch + c];
    } else if (!strcmp(cmd, "close")) {
        vo_mex_check(vo_ingest_close(pool_of(prhs[1])), "vo_ingest_close");
    } else {
        mexErrMsgIdAndTxt("vo:args", "unknown command %s", cmd);
    }
}
