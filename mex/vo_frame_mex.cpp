// Device-resident VisualOdometry.processFrame (modules/VisualOdometry.m:264-346) behind one gateway:
//   h = vo_frame_mex('create', opts, [H W], K)          opts: the properties of :22-37 (+ Seed), K = IntrinsicMatrix
//   vo_frame_mex('set_state', h, state, prev_img)       state: the struct of :308-315 / :338-345 (poses as a cell)
//   [pose, status, counts] = vo_frame_mex('process', h, img)         img: H x W uint8, or a uint64 device handle
//   [pose, status, counts] = vo_frame_mex('process', h, dimg, dnext) (from vo_ingest_mex('next', ...)); dnext: the
//                                                                     handle of the frame after it (pool opened with
//                                                                     hold = 2), announced with vo_frame_announce
//   state = vo_frame_mex('get_state', h)
//   vo_frame_mex('destroy', h)
// The loop works zero-based and row-major: points lose 1, the principal point loses 1, images and poses are
// transposed here (a MATLAB H x W array is W columns of H bytes).
#include <string.h>

#include <vector>

#include "vo_mex_common.h"

static vo_frame_state* handle_of(const mxArray* a) { return (vo_frame_state*)(uintptr_t)(*(uint64_t*)mxGetData(a)); }
static int g_H = 0, g_W = 0;

static void rows_from_matlab(const mxArray* a, int cols, double shift, std::vector<double>* out) {   // N x cols col-major -> row-major
    const int n = a ? (int)mxGetM(a) : 0;
    out->assign((size_t)n * cols, 0.0);
    if (!n) return;
    const bool dbl = mxIsDouble(a);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < cols; ++c)
            (*out)[(size_t)i * cols + c] = (dbl ? mxGetPr(a)[(size_t)c * n + i] : (double)((float*)mxGetData(a))[(size_t)c * n + i]) + shift;
}
static mxArray* rows_to_matlab(const std::vector<double>& v, int n, int cols, double shift) {
    mxArray* a = mxCreateDoubleMatrix(n, cols, mxREAL);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < cols; ++c) mxGetPr(a)[(size_t)c * n + i] = v[(size_t)i * cols + c] + shift;
    return a;
}
static void image_rows(const mxArray* img, std::vector<uint8_t>* out) {   // column-major H x W -> H rows of W bytes
    if (!mxIsUint8(img) || (int)mxGetM(img) != g_H || (int)mxGetN(img) != g_W) mexErrMsgIdAndTxt("vo:args", "image must be H x W uint8");
    const uint8_t* s = (const uint8_t*)mxGetData(img);
    out->resize((size_t)g_H * g_W);
    for (int x = 0; x < g_W; ++x)
        for (int y = 0; y < g_H; ++y) (*out)[(size_t)y * g_W + x] = s[(size_t)x * g_H + y];
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    char cmd[32];
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("vo:args", "first argument: command string");
    vo_ctx* ctx = vo_mex_ctx();
    if (!strcmp(cmd, "create")) {
        const mxArray* o = prhs[1];
        vo_frame_opts f;
        memset(&f, 0, sizeof f);
        f.max_landmarks = (int)vo_opt(o, "maxNumLandmarks", 300);            // :24
        f.max_candidates = (int)vo_opt(o, "maxCandidates", 2048);
        f.max_temporal_recall = (int)vo_opt(o, "maxTemporalRecall", 10);     // :23
        f.angular_threshold_deg = vo_opt(o, "angularThreshold", 1.0);        // :22
        f.max_reproj_err = vo_opt(o, "maxReprojectionError", 3.0);           // :25
        f.klt.levels = 4; f.klt.max_bidir_err = 2; f.klt.block_h = f.klt.block_w = 41; f.klt.max_iters = 50; f.klt.eps = 0.01;   // :33-37
        f.pose.num_hyp = 5000; f.pose.max_reproj_err = 2; f.pose.confidence = 0.99; f.pose.mode = 1; f.pose.min_inliers = 4;     // :285-289
        f.pose.seed = (uint64_t)vo_opt(o, "Seed", 1023);
        f.sel.min_distance = 7; f.sel.min_quality = 0.001; f.sel.filter_size = 5; f.sel.keep = 100; f.sel.scorer = VO_SCORER_MINEIGEN;   // :332-335
        g_H = (int)mxGetPr(prhs[2])[0]; g_W = (int)mxGetPr(prhs[2])[1];
        double K[9];
        memcpy(K, mxGetPr(prhs[3]), sizeof K);   // IntrinsicMatrix column-major = K row-major
        K[2] -= 1.0; K[5] -= 1.0;                // zero-based pixel centres
        vo_frame_state* st = nullptr;
        vo_mex_check(vo_frame_create(ctx, &f, g_W, g_H, K, &st), "vo_frame_create");
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *(uint64_t*)mxGetData(plhs[0]) = (uint64_t)(uintptr_t)st;
    } else if (!strcmp(cmd, "set_state")) {
        vo_frame_state* st = handle_of(prhs[1]);
        const mxArray* s = prhs[2];
        std::vector<double> L, kp, rep, ck, cf, cp;
        rows_from_matlab(mxGetField(s, 0, "landmarks"), 3, 0.0, &L);
        rows_from_matlab(mxGetField(s, 0, "keypoints"), 2, -1.0, &kp);
        const int N = (int)(kp.size() / 2);
        const mxArray* re = mxGetField(s, 0, "reproError");
        rep.assign(N, 0.0);
        if (re) for (int i = 0; i < N; ++i) rep[i] = mxGetPr(re)[i];
        rows_from_matlab(mxGetField(s, 0, "candidate_keypoints"), 2, -1.0, &ck);
        rows_from_matlab(mxGetField(s, 0, "candidate_first_keypoints"), 2, -1.0, &cf);
        const int C = (int)(ck.size() / 2);
        const mxArray* poses = mxGetField(s, 0, "candidate_first_poses");   // 1 x C cell of 4 x 3
        const mxArray* ti = mxGetField(s, 0, "candidate_time_indxs");
        cp.assign((size_t)C * 12, 0.0);
        std::vector<int32_t> ct(C, -1);
        for (int i = 0; i < C; ++i) {
            const double* p = mxGetPr(mxGetCell(poses, i));
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 3; ++c) cp[(size_t)i * 12 + r * 3 + c] = p[c * 4 + r];
            ct[i] = (int32_t)mxGetPr(ti)[i];
        }
        vo_mex_check(vo_frame_set_state(ctx, st, L.data(), kp.data(), rep.data(), N, ck.data(), cf.data(), cp.data(), ct.data(), C, 0),
                     "vo_frame_set_state");
        std::vector<uint8_t> img;
        image_rows(prhs[3], &img);
        vo_mex_check(vo_frame_set_image(ctx, st, img.data(), 0), "vo_frame_set_image");
    } else if (!strcmp(cmd, "process") || !strcmp(cmd, "submit") || !strcmp(cmd, "collect")) {
        // 'process' = 'submit' + 'collect'.  A loop that calls 'submit' for frame k+1 before 'collect' for frame k keeps the
        // GPU fed while MATLAB runs (up to VO_FRAME_MAX_PENDING frames may be out): [pose, status, counts] come from 'collect'.
        vo_frame_state* st = handle_of(prhs[1]);
        double pose[12];
        vo_frame_info info;
        if (strcmp(cmd, "collect")) {
            if (mxIsUint64(prhs[2])) {   // a device image of the decode pool: row-major already
                if (nrhs > 3 && mxIsUint64(prhs[3]) && *(uint64_t*)mxGetData(prhs[3]))   // look-ahead: the frame after this one
                    vo_mex_check(vo_frame_announce(ctx, st, (const uint8_t*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[3])), 1), "vo_frame_announce");
                vo_mex_check(vo_frame_submit(ctx, st, (const uint8_t*)(uintptr_t)(*(uint64_t*)mxGetData(prhs[2])), 1), "vo_frame_submit");
            } else {
                std::vector<uint8_t> img;
                image_rows(prhs[2], &img);
                vo_mex_check(vo_frame_submit(ctx, st, img.data(), 0), "vo_frame_submit");   // the host image is copied during the call
            }
            if (!strcmp(cmd, "submit")) return;
        }
        vo_mex_check(vo_frame_collect(ctx, st, pose, &info), "vo_frame_collect");
        plhs[0] = mxCreateDoubleMatrix(4, 3, mxREAL);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 3; ++c) mxGetPr(plhs[0])[c * 4 + r] = pose[r * 3 + c];
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar((double)info.status);
        if (nlhs > 2) {
            const char* names[] = {"n_landmarks", "n_candidates", "n_tracked", "n_inliers", "n_promoted", "n_new"};
            const int32_t vals[] = {info.n_landmarks, info.n_candidates, info.n_tracked, info.n_inliers, info.n_promoted, info.n_new};
            plhs[2] = mxCreateStructMatrix(1, 1, 6, names);
            for (int k = 0; k < 6; ++k) mxSetField(plhs[2], 0, names[k], mxCreateDoubleScalar((double)vals[k]));
        }
    } else if (!strcmp(cmd, "get_state")) {
        vo_frame_state* st = handle_of(prhs[1]);
        int N = 0, C = 0;
        vo_mex_check(vo_frame_get_counts(ctx, st, &N, &C), "vo_frame_get_counts");
        std::vector<double> L((size_t)N * 3), kp((size_t)N * 2), rep(N), ck((size_t)C * 2), cf((size_t)C * 2), cp((size_t)C * 12);
        std::vector<int32_t> ct(C);
        vo_mex_check(vo_frame_get_state(ctx, st, L.data(), kp.data(), rep.data(), ck.data(), cf.data(), cp.data(), ct.data()), "vo_frame_get_state");
        const char* names[] = {"landmarks", "keypoints", "reproError", "candidate_keypoints", "candidate_first_keypoints",
                               "candidate_first_poses", "candidate_time_indxs"};
        plhs[0] = mxCreateStructMatrix(1, 1, 7, names);
        mxSetField(plhs[0], 0, names[0], rows_to_matlab(L, N, 3, 0.0));
        mxSetField(plhs[0], 0, names[1], rows_to_matlab(kp, N, 2, 1.0));
        mxSetField(plhs[0], 0, names[2], rows_to_matlab(rep, N, 1, 0.0));
        mxSetField(plhs[0], 0, names[3], rows_to_matlab(ck, C, 2, 1.0));
        mxSetField(plhs[0], 0, names[4], rows_to_matlab(cf, C, 2, 1.0));
        mxArray* cell = mxCreateCellMatrix(1, C);
        mxArray* tidx = mxCreateDoubleMatrix(1, C, mxREAL);
        for (int i = 0; i < C; ++i) {
            mxArray* p = mxCreateDoubleMatrix(4, 3, mxREAL);
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 3; ++c) mxGetPr(p)[c * 4 + r] = cp[(size_t)i * 12 + r * 3 + c];
            mxSetCell(cell, i, p);
            mxGetPr(tidx)[i] = (double)ct[i];
        }
        mxSetField(plhs[0], 0, names[5], cell);
        mxSetField(plhs[0], 0, names[6], tidx);
    } else if (!strcmp(cmd, "destroy")) {
        vo_mex_check(vo_frame_destroy(ctx, handle_of(prhs[1])), "vo_frame_destroy");
    } else {
        mexErrMsgIdAndTxt("vo:args", "unknown command %s", cmd);
    }
}
