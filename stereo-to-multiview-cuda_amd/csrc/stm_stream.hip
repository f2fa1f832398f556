// stm_stream.hip -- pipelined side-by-side frame sequence processor.
//
// The reference's video loop (video_io.cpp:144-165) calls adcensus_stm once per decoded frame; every call
// uploads the frame, computes, downloads three results and only then returns, so PCIe transfers and compute
// never overlap.  This front end keeps the same per-frame contract (one SBS frame in; disp_l, disp_r and the
// interlaced frame out, in submission order) but runs an upload stream, a download stream and one COMPUTE stream +
// private workspace per buffer slot over double-buffered pinned / device buffers: while frame k computes, frame k+1 is
// uploading and frame k-1 is downloading, and (round 3) the two frames in flight also overlap ON the GPU: the
// latency-bound tail of frame k (region voting, filters, view synthesis) shares the chip with the issue-bound aggregation
// of frame k+1 (bench.py's rate_two_in_flight measures that overlap for device-resident frames).  STM_STREAM_OVERLAP=0: one
// compute stream and workspace for both slots, as in round 2.
#include "stm_common.h"
#include "../../include/stm_hip.h"

#include <stdlib.h>
#include <string.h>

namespace {

struct Slot {
    u8 *h_in = nullptr, *d_in = nullptr, *d_out = nullptr, *h_out = nullptr;
    float *d_dl = nullptr, *d_dr = nullptr, *h_dl = nullptr, *h_dr = nullptr;
    hipEvent_t ev_in, ev_done, ev_out;
    bool busy = false;
    hipStream_t s_compute = nullptr; // this slot's compute stream and private workspace (shared by both slots when overlap is off)
    void *ws = nullptr;              // private: the addresses baked into the slot's graph stay valid
    // the slot's frame pipeline as a captured graph: a frame is ~30 launches with fixed arguments (the slot's buffers,
    // the stream's private workspace), replayed with one hipGraphLaunch
    hipGraphExec_t gexec = nullptr;
    void *g_ws_base = nullptr;
    size_t g_ws_cap = 0;
    int eager_runs = 0;
};

struct FrameStream {
    int H, Wsbs, W, Hout, Wout, E, N, D, zd, usd, lsd, thresh_s;
    float angle, ad, ce, ucd, lcd, thresh_h;
    size_t in_sz, out_sz, hw;
    int dev = 0; // the device the stream was created on: submit / collect switch to it (and back) if the caller's differs
    hipStream_t s_in, s_out;
    bool overlap = true; // two frames in flight on the GPU (a compute stream + workspace per slot)
    bool use_graph = true;
    Slot slot[2];
    long submitted = 0, collected = 0;
};

} // namespace

extern "C" {

void *stm_stream_create(int num_rows, int num_cols_sbs, int num_cols, int num_rows_out, int num_cols_out, int elem_sz,
                        int num_views, float angle, int num_disp, int zero_disp, float ad_coeff, float census_coeff,
                        float ucd, float lcd, int usd, int lsd, int thresh_s, float thresh_h)
{
    FrameStream *f = new FrameStream();
    f->H = num_rows; f->Wsbs = num_cols_sbs; f->W = num_cols; f->Hout = num_rows_out; f->Wout = num_cols_out; f->E = elem_sz;
    f->N = num_views; f->angle = angle; f->D = num_disp; f->zd = zero_disp; f->ad = ad_coeff; f->ce = census_coeff;
    f->ucd = ucd; f->lcd = lcd; f->usd = usd; f->lsd = lsd; f->thresh_s = thresh_s; f->thresh_h = thresh_h;
    f->in_sz = (size_t)num_rows * num_cols_sbs * elem_sz;
    f->out_sz = (size_t)num_rows_out * num_cols_out * elem_sz;
    f->hw = (size_t)num_rows * num_cols;
    STM_CHECK(hipGetDevice(&f->dev));
    STM_CHECK(hipStreamCreateWithFlags(&f->s_in, hipStreamNonBlocking));
    STM_CHECK(hipStreamCreateWithFlags(&f->s_out, hipStreamNonBlocking));
    const char *g = getenv("STM_STREAM_GRAPH"); // STM_STREAM_GRAPH=0: always launch kernel by kernel
    f->use_graph = !(g && g[0] == '0');
    const char *o = getenv("STM_STREAM_OVERLAP"); // STM_STREAM_OVERLAP=0: one compute stream + workspace for both slots
    f->overlap = !(o && o[0] == '0');
    for (int i = 0; i < 2; ++i) {
        Slot &s = f->slot[i];
        if (i == 0 || f->overlap) {
            STM_CHECK(hipStreamCreateWithFlags(&s.s_compute, hipStreamNonBlocking));
            s.ws = stm::ws_private_create();
        } else {
            s.s_compute = f->slot[0].s_compute;
            s.ws = f->slot[0].ws;
        }
        STM_CHECK(hipHostMalloc((void **)&s.h_in, f->in_sz, hipHostMallocDefault));
        STM_CHECK(hipHostMalloc((void **)&s.h_out, f->out_sz, hipHostMallocDefault));
        STM_CHECK(hipHostMalloc((void **)&s.h_dl, f->hw * 4, hipHostMallocDefault));
        STM_CHECK(hipHostMalloc((void **)&s.h_dr, f->hw * 4, hipHostMallocDefault));
        STM_CHECK(hipMalloc((void **)&s.d_in, f->in_sz));
        STM_CHECK(hipMalloc((void **)&s.d_out, f->out_sz));
        STM_CHECK(hipMalloc((void **)&s.d_dl, f->hw * 4));
        STM_CHECK(hipMalloc((void **)&s.d_dr, f->hw * 4));
        STM_CHECK(hipMemsetAsync(s.d_out, 0, f->out_sz, s.s_compute)); // ordered before the first frame's writes
        STM_CHECK(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        STM_CHECK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
        STM_CHECK(hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
    }
    return f;
}

// Stage frame `submitted`; at most two frames may be in flight (collect the older one first).
// Returns the frame's index, or -1 when both slots are still uncollected.
long stm_stream_submit(void *h, const unsigned char *img_sbs)
{
    FrameStream *f = (FrameStream *)h;
    Slot &s = f->slot[f->submitted & 1];
    if (s.busy) return -1;
    stm::clear_failed();
    int caller_dev = f->dev;
    STM_CHECK(hipGetDevice(&caller_dev));
    if (caller_dev != f->dev) STM_CHECK(hipSetDevice(f->dev));
    // the caller's buffer is free again when this returns (as with adcensus_stm); a frame that was written straight into the
    // slot's pinned buffer (stm_stream_input_buffer) needs no copy
    if (img_sbs && img_sbs != s.h_in) memcpy(s.h_in, img_sbs, f->in_sz);
    STM_CHECK(hipMemcpyAsync(s.d_in, s.h_in, f->in_sz, hipMemcpyHostToDevice, f->s_in));
    STM_CHECK(hipEventRecord(s.ev_in, f->s_in));
    STM_CHECK(hipStreamWaitEvent(s.s_compute, s.ev_in, 0));
    void *prev = stm_get_stream();
    stm_set_stream(s.s_compute);
    stm::ws_private_bind(s.ws);
    auto pipeline = [&]() {
        stm::ApiNest nest; // a failed upload above must survive the nested call's argument screen
        stm_d_adcensus_stm(s.d_in, s.d_dl, s.d_dr, s.d_out, f->H, f->Wsbs, f->W, f->Hout, f->Wout, f->E, f->N, f->angle, f->D,
                           f->zd, f->ad, f->ce, f->ucd, f->lcd, f->usd, f->lsd, f->thresh_s, f->thresh_h, 3);
    };
    void *wb = nullptr;
    size_t wc = 0;
    stm::ws_identity(&wb, &wc);
    if (s.gexec && (wb != s.g_ws_base || wc != s.g_ws_cap)) { // cannot happen with a private workspace; never replay stale addresses
        STM_CHECK(hipGraphExecDestroy(s.gexec));
        s.gexec = nullptr;
    }
    if (s.gexec) {
        STM_CHECK(hipGraphLaunch(s.gexec, s.s_compute));
    } else if (f->use_graph && s.eager_runs >= 1 && !stm::prof_enabled()) {
        // the slot's first frame ran eagerly (it sized the workspace, built the lookup tables, raised the LDS limits), so
        // nothing in here allocates or synchronises: capture this frame's launches, then run the capture
        hipGraph_t graph = nullptr;
        STM_CHECK(hipStreamBeginCapture(s.s_compute, hipStreamCaptureModeThreadLocal));
        pipeline();
        const bool capture_failed = stm::failed(); // error mode 1: something inside the capture recorded an error
        STM_CHECK(hipStreamEndCapture(s.s_compute, &graph)); // always leave capture mode
        if (capture_failed || !graph) {
            if (graph) STM_CHECK(hipGraphDestroy(graph));
            f->use_graph = false; // stay eager from now on
            stm::ws_private_bind(nullptr);
            stm_set_stream(prev);
            if (caller_dev != f->dev) STM_CHECK(hipSetDevice(caller_dev));
            return -1;
        }
        STM_CHECK(hipGraphInstantiate(&s.gexec, graph, nullptr, nullptr, 0));
        STM_CHECK(hipGraphDestroy(graph));
        stm::ws_identity(&s.g_ws_base, &s.g_ws_cap);
        STM_CHECK(hipGraphLaunch(s.gexec, s.s_compute));
    } else {
        pipeline();
        ++s.eager_runs;
    }
    stm::ws_private_bind(nullptr);
    stm_set_stream(prev);
    STM_CHECK(hipEventRecord(s.ev_done, s.s_compute));
    STM_CHECK(hipStreamWaitEvent(f->s_out, s.ev_done, 0));
    STM_CHECK(hipMemcpyAsync(s.h_dl, s.d_dl, f->hw * 4, hipMemcpyDeviceToHost, f->s_out));
    STM_CHECK(hipMemcpyAsync(s.h_dr, s.d_dr, f->hw * 4, hipMemcpyDeviceToHost, f->s_out));
    STM_CHECK(hipMemcpyAsync(s.h_out, s.d_out, f->out_sz, hipMemcpyDeviceToHost, f->s_out));
    STM_CHECK(hipEventRecord(s.ev_out, f->s_out));
    if (caller_dev != f->dev) STM_CHECK(hipSetDevice(caller_dev));
    if (stm::failed()) return -1; // error mode 1: the frame was not (completely) enqueued
    s.busy = true;
    return f->submitted++;
}

// Blocks until the oldest uncollected frame is complete and copies its results out (any pointer may be NULL).
// Returns that frame's index, or -1 when nothing is pending.
long stm_stream_collect(void *h, float *disp_l, float *disp_r, unsigned char *interlaced)
{
    FrameStream *f = (FrameStream *)h;
    if (f->collected >= f->submitted) return -1;
    Slot &s = f->slot[f->collected & 1];
    STM_CHECK(hipEventSynchronize(s.ev_out));
    if (disp_l) memcpy(disp_l, s.h_dl, f->hw * 4);
    if (disp_r) memcpy(disp_r, s.h_dr, f->hw * 4);
    if (interlaced) memcpy(interlaced, s.h_out, f->out_sz);
    s.busy = false;
    return f->collected++;
}

// Zero-copy variants: at 1080p the two host-side copies (12 MB in, 23 MB out) take longer than the frame does on the GPU.
// The pinned input buffer of the slot the NEXT submit will use: decode / write the frame into it, then call
// stm_stream_submit(stream, that pointer) (or NULL).  NULL while that slot is still uncollected.
unsigned char *stm_stream_input_buffer(void *h)
{
    FrameStream *f = (FrameStream *)h;
    Slot &s = f->slot[f->submitted & 1];
    return s.busy ? nullptr : s.h_in;
}

// Waits for the oldest uncollected frame and hands out pointers to its pinned result buffers instead of copying them; the
// pointers stay valid until the frame after the next one is submitted (its slot is reused then).  Returns the index or -1.
long stm_stream_collect_view(void *h, const float **disp_l, const float **disp_r, const unsigned char **interlaced)
{
    FrameStream *f = (FrameStream *)h;
    if (f->collected >= f->submitted) return -1;
    Slot &s = f->slot[f->collected & 1];
    STM_CHECK(hipEventSynchronize(s.ev_out));
    if (disp_l) *disp_l = s.h_dl;
    if (disp_r) *disp_r = s.h_dr;
    if (interlaced) *interlaced = s.h_out;
    s.busy = false;
    return f->collected++;
}

void stm_stream_destroy(void *h)
{
    FrameStream *f = (FrameStream *)h;
    if (!f) return;
    STM_CHECK(hipStreamSynchronize(f->s_in));
    for (Slot &s : f->slot) STM_CHECK(hipStreamSynchronize(s.s_compute));
    STM_CHECK(hipStreamSynchronize(f->s_out));
    for (Slot &s : f->slot) {
        STM_CHECK(hipHostFree(s.h_in)); STM_CHECK(hipHostFree(s.h_out)); STM_CHECK(hipHostFree(s.h_dl)); STM_CHECK(hipHostFree(s.h_dr));
        STM_CHECK(hipFree(s.d_in)); STM_CHECK(hipFree(s.d_out)); STM_CHECK(hipFree(s.d_dl)); STM_CHECK(hipFree(s.d_dr));
        STM_CHECK(hipEventDestroy(s.ev_in)); STM_CHECK(hipEventDestroy(s.ev_done)); STM_CHECK(hipEventDestroy(s.ev_out));
        if (s.gexec) STM_CHECK(hipGraphExecDestroy(s.gexec));
    }
    stm::ws_private_destroy(f->slot[0].ws);
    STM_CHECK(hipStreamDestroy(f->slot[0].s_compute));
    if (f->overlap) {
        stm::ws_private_destroy(f->slot[1].ws);
        STM_CHECK(hipStreamDestroy(f->slot[1].s_compute));
    }
    STM_CHECK(hipStreamDestroy(f->s_in)); STM_CHECK(hipStreamDestroy(f->s_out));
    delete f;
}

} // extern "C"
