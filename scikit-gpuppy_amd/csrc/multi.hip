// multi.hip -- the sharded path behind the C-ABI: ONE host process driving R devices of one node.
//
// The reference factors on one host (skgpuppy/Covariance.py:179, scipy.linalg.inv) and predicts with two dense products
// (skgpuppy/GaussianProcess.py:75-78); it has no distributed code.  skgpuppy_amd/distributed.py is the one-process-per-GPU form of the
// sharded path (torch.distributed / RCCL); this file is the same schedule for a caller WITHOUT Python or a process launcher -- a non-Python
// binding reaches e1-e4 through gpx_multi_*:
//   e1  K-build: every device assembles only the outer panels (1024 columns) it owns, block-cyclically; no exchange
//   e2  panel Cholesky with look-ahead: the owner of panel p + 1 applies panel p to it and factors it (gpx_dev_chol_panel_split: chain on
//       the device's side stream, the next square's rows and the rows below on two more streams, column by column behind it) while every device's main stream applies panel p to the
//       other panels it owns; a finished panel is ONE buffer (rows below | square | inverted diagonal blocks | diagonal) that travels in
//       TWO parts -- HEAD: the rows of the next panel's diagonal square, all the next owner needs to start its chain; TAIL: the rest -- by
//       peer-to-peer copies (hipMemcpyPeerAsync over xGMI: R - 1 direct sends from the owner per part, posted on the RECEIVER's head / far
//       stream) into one of three staging slots per device; events order every reuse; nothing blocks the host until the end
//   e3  every device ends with the complete factor (gpx_adopt_factor); estimate_many shards the QUERIES, one host thread per device
//   e4  propagate_GA: the d + 1 right-hand sides [C, J_1..J_d] are dealt to the devices (gpx_propagate_approx_rhs), the 4 + 2 d partial sums
//       meet on the host (the "all-reduce" of one process) -- UncertaintyPropagation.py:397-479
// A non-positive pivot anywhere is answered by ONE collective retry on K + 1e-5 I (Covariance.py:180-185); a timed-out in-kernel hand-off
// (GPX_INFO_STALLED) is an error, never jitter.  Devices may repeat in the list (two logical ranks on one GPU): that is how the path is
// tested on a one-GPU box (tests/test_gpu_parity.py::test_multi_device_abi_on_one_gpu).  Unmeasured on more than one GPU.
#include <algorithm>
#include <cmath>
#include <map>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr int64_t PB = CHOL_PANEL_COLS / TILE;   // blocks per outer panel
constexpr int NSLOTS = 3;                        // staging buffers per device: panel p's message lives in slot p % 3

struct Operand { const double *ptr = nullptr; int64_t ld = 0, first = 0; };

struct MDev {
    int dev = 0;
    hipStream_t main = nullptr, side = nullptr, copy = nullptr, head = nullptr, far = nullptr;
    double *xw = nullptr, *L = nullptr, *Dinv = nullptr, *diag = nullptr, *stage[NSLOTS] = {nullptr, nullptr, nullptr};
    int *info = nullptr;
    gpx_handle *h = nullptr;
    hipEvent_t ev_built = nullptr;
    std::vector<hipEvent_t> ev_head;                    // the rows of panel p + 1's square (the message's HEAD) are on this device
    std::vector<hipEvent_t> ev_tail;                    // panel p's message is complete on this device (owner: solved; others: arrived)
    std::vector<hipEvent_t> ev_look;                    // the main stream has applied every panel but the last one to panel p
    std::vector<std::vector<hipEvent_t>> readers;       // events behind every read of panel p's message buffer on this device
    std::vector<Operand> operand;
    std::vector<hipEvent_t> all_events;
};

}   // namespace

struct gpx_multi {
    int64_t n = 0, npad = 0, nblk = 0, npanels = 0;
    int d = 0;
    double theta[GPX_MAX_D + 2];
    double v = 0, vt = 0, jitter = 0;
    std::vector<MDev> devs;
};

namespace {

int set_dev(const MDev &m)
{
    GPX_TRY(gpx_set_device(m.dev));          // the library's thread-local choice (every gpx_* call re-selects it)
    GPX_HIP(hipSetDevice(m.dev));
    return 0;
}

int new_event(MDev &m, hipStream_t on, hipEvent_t *out)
{
    hipEvent_t e = nullptr;
    GPX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    m.all_events.push_back(e);
    GPX_HIP(hipEventRecord(e, on));
    *out = e;
    return 0;
}

struct Geom { int64_t b0, b1, c0, w, rows; };
Geom geom(const gpx_multi *g, int64_t p)
{
    Geom q;
    q.b0 = p * PB;
    q.b1 = std::min<int64_t>(q.b0 + PB, g->nblk);
    q.c0 = q.b0 * TILE;
    q.w = (q.b1 - q.b0) * TILE;
    q.rows = g->npad - q.c0;
    return q;
}
int64_t message_elems(const gpx_multi *g, int64_t p)
{
    const Geom q = geom(g, p);
    return q.rows * q.w + (q.b1 - q.b0) * (int64_t)TILE * TILE + q.w;   // rows below + square | inverted blocks | diagonal
}
int owner(const gpx_multi *g, int64_t p) { return (int)(p % (int64_t)g->devs.size()); }
// A panel's message travels in two parts when there is more than one rank: the HEAD -- its first rows below the square, i.e. the rows of
// the NEXT panel's diagonal square, all the next owner needs to start its chain -- and the TAIL (the rows further down, the square, the
// inverted blocks, the diagonal).  head_rows(p): rows of the head (0: one part).
// Default: split where the ranks sit on DIFFERENT devices (as skgpuppy_amd/distributed.py does above one rank), whole otherwise -- a single
// rank gains nothing from the extra row slice (C3 30.3 against 29.3 ms), and logical ranks that SHARE one GPU (the test rehearsal) are
// faster with whole messages (41 against 51 ms for two at C3: ten streams on one chip).  GPX_PANEL_MESSAGE=split / whole overrides.
bool split_message(const gpx_multi *g)
{
    static const int env = [] { const char *e = getenv("GPX_PANEL_MESSAGE"); return !e ? 0 : (std::string(e) == "split" ? 1 : (std::string(e) == "whole" ? 2 : 0)); }();
    if (env) return env == 1;
    for (const MDev &m : g->devs)
        if (m.dev != g->devs[0].dev) return true;      // ranks with a chip each
    return false;                                      // one rank, or logical ranks that share one GPU
}
int64_t head_rows(const gpx_multi *g, int64_t p)
{
    if (!split_message(g) || p + 1 >= g->npanels) return 0;
    return geom(g, p + 1).w;
}

// message buffer of panel p on device m: lower [rows - w, w] | square [w, w] | dinv | diag
struct Views { double *lower, *square, *dinv, *diag; };
Views views(const gpx_multi *g, MDev &m, int64_t p)
{
    const Geom q = geom(g, p);
    double *buf = m.stage[p % NSLOTS];
    Views v;
    v.lower = buf;
    v.square = buf + (q.rows - q.w) * q.w;
    v.dinv = v.square + q.w * q.w;
    v.diag = v.dinv + (q.b1 - q.b0) * (int64_t)TILE * TILE;
    return v;
}

int wait_slot_free(MDev &m, int64_t p, hipStream_t on)
{
    if (p - NSLOTS < 0) return 0;
    for (hipEvent_t e : m.readers[(size_t)(p - NSLOTS)]) GPX_HIP(hipStreamWaitEvent(on, e, 0));
    return 0;
}

// Gram columns of panel p, rows from its diagonal down (+vt on the diagonal, identity padding) on the device's main stream
int build_panel(gpx_multi *g, MDev &m, int64_t p)
{
    const Geom q = geom(g, p);
    GPX_TRY(set_dev(m));
    double *C = m.L + q.c0 * g->npad + q.c0;
    if (q.c0 >= g->n) {   // panel entirely in the padding: identity (never happens for npad = round_up(n, 128); kept for safety)
        GPX_HIP(hipMemset2DAsync(C, sizeof(double) * g->npad, 0, sizeof(double) * q.w, q.rows, m.main));
        return 0;
    }
    const double *xi = m.xw + q.c0 * g->d;
    return gpx_dev_gram_scaled(xi, g->n - q.c0, xi, std::min<int64_t>(q.c0 + q.w, g->n) - q.c0, g->d, g->v, g->vt + g->jitter, 0, 1, C, g->npad, q.rows,
                               q.w, m.main);
}

// the owner's step: (update with `prev`,) factor panel p -- chain on the side stream, the next square's rows on the head stream, the rows
// further down on the far stream, both column by column behind the chain (gpx_dev_chol_panel_split) -- and pack the message's two parts,
// each on the stream that solved its rows
int factor_panel(gpx_multi *g, MDev &m, int64_t p, int64_t prev)
{
    const Geom q = geom(g, p);
    const int64_t hr = head_rows(g, p), below = q.rows - q.w;
    GPX_TRY(set_dev(m));
    const bool split = split_message(g);
    // the streams of the step: chain, the next square's rows, the rows further down -- ONE stream for all three where the message is whole
    hipStream_t s_head = split ? m.head : m.side, s_far = split ? m.far : m.side;
    std::vector<hipStream_t> three = split ? std::vector<hipStream_t>{m.side, m.head, m.far} : std::vector<hipStream_t>{m.side};
    for (hipStream_t st : three) GPX_HIP(hipStreamWaitEvent(st, m.ev_built, 0));   // this device's panels are assembled (main stream)
    const double *P = nullptr;
    int64_t ldp = 0, kp = 0;
    if (prev >= 0) {
        // the chain needs the HEAD of `prev` (this panel's square rows of it) and the main stream's earlier updates of this panel; the rows
        // below the square need prev's TAIL as well
        GPX_HIP(hipStreamWaitEvent(m.side, m.ev_head[(size_t)prev], 0));
        GPX_HIP(hipStreamWaitEvent(s_head, m.ev_tail[(size_t)prev], 0));
        GPX_HIP(hipStreamWaitEvent(s_far, m.ev_tail[(size_t)prev], 0));
        if (m.ev_look[(size_t)p])
            for (hipStream_t st : three) GPX_HIP(hipStreamWaitEvent(st, m.ev_look[(size_t)p], 0));   // main has applied the panels before prev
        const Operand &o = m.operand[(size_t)prev];
        P = o.ptr + (q.c0 - o.first) * o.ld;
        ldp = o.ld;
        kp = geom(g, prev).w;
    }
    if (split) {
        GPX_TRY(gpx_dev_chol_panel_split(m.L, g->npad, g->nblk, q.b0, q.b1, hr / TILE, P, ldp, kp, m.Dinv, m.diag, m.info, m.side, m.head, m.far));
        hipEvent_t chain_done;
        GPX_TRY(new_event(m, m.side, &chain_done));
        GPX_HIP(hipStreamWaitEvent(m.far, chain_done, 0));      // (square, inverted blocks and diagonal are the chain's: packed with the tail)
    } else if (P) {
        GPX_TRY(gpx_dev_chol_panel_next(m.L, g->npad, g->nblk, q.b0, q.b1, P, ldp, kp, m.Dinv, m.diag, m.info, m.side));   // (one internal row stream, joined back)
    } else {
        GPX_TRY(gpx_dev_chol_panel(m.L, g->npad, g->nblk, q.b0, q.b1, m.Dinv, m.diag, m.info, m.side));
    }
    if (prev >= 0)
        for (hipStream_t st : three) {                           // the step's reads of prev's message buffer
            hipEvent_t e;
            GPX_TRY(new_event(m, st, &e));
            m.readers[(size_t)prev].push_back(e);
        }
    if (g->devs.size() > 1) {
        // pack each part behind ITS rows' solves, once the slot's previous tenant has been read everywhere it is read on this device
        const Views v = views(g, m, p);
        if (hr > 0) {
            GPX_TRY(wait_slot_free(m, p, s_head));
            GPX_HIP(hipMemcpy2DAsync(v.lower, sizeof(double) * q.w, m.L + (q.c0 + q.w) * g->npad + q.c0, sizeof(double) * g->npad, sizeof(double) * q.w, hr,
                                     hipMemcpyDeviceToDevice, s_head));
        }
        GPX_TRY(wait_slot_free(m, p, s_far));
        if (below > hr)
            GPX_HIP(hipMemcpy2DAsync(v.lower + hr * q.w, sizeof(double) * q.w, m.L + (q.c0 + q.w + hr) * g->npad + q.c0, sizeof(double) * g->npad,
                                     sizeof(double) * q.w, below - hr, hipMemcpyDeviceToDevice, s_far));
        GPX_HIP(hipMemcpy2DAsync(v.square, sizeof(double) * q.w, m.L + q.c0 * g->npad + q.c0, sizeof(double) * g->npad, sizeof(double) * q.w, q.w,
                                 hipMemcpyDeviceToDevice, s_far));
        GPX_HIP(hipMemcpyAsync(v.dinv, m.Dinv + q.b0 * (int64_t)TILE * TILE, sizeof(double) * (q.b1 - q.b0) * TILE * TILE, hipMemcpyDeviceToDevice, s_far));
        GPX_HIP(hipMemcpyAsync(v.diag, m.diag + q.c0, sizeof(double) * q.w, hipMemcpyDeviceToDevice, s_far));
    }
    GPX_TRY(new_event(m, s_far, &m.ev_tail[(size_t)p]));
    if (hr > 0) GPX_TRY(new_event(m, s_head, &m.ev_head[(size_t)p]));
    else m.ev_head[(size_t)p] = m.ev_tail[(size_t)p];
    // the owner's own updates read the panel in place
    m.operand[(size_t)p].ptr = m.L + (q.c0 + q.w) * g->npad + q.c0;
    m.operand[(size_t)p].ld = g->npad;
    m.operand[(size_t)p].first = q.c0 + q.w;
    return 0;
}

// panel p from its owner to device r: the head on the receiver's head stream, the tail on its far stream (two peer copies, the head first),
// then -- copy stream -- into r's own L / Dinv / diag for the complete factor
int receive_panel(gpx_multi *g, MDev &src, MDev &dst, int64_t p)
{
    const Geom q = geom(g, p);
    const int64_t hr = head_rows(g, p), below = q.rows - q.w;
    GPX_TRY(set_dev(dst));
    double *dbuf = dst.stage[p % NSLOTS];
    const double *sbuf = src.stage[p % NSLOTS];
    const size_t head_elems = (size_t)(hr * q.w), all_elems = (size_t)message_elems(g, p);
    hipStream_t r_far = split_message(g) ? dst.far : dst.copy;                   // (whole message: the receiver's copy stream carries the transfer)
    if (hr > 0) {
        GPX_TRY(wait_slot_free(dst, p, dst.head));
        GPX_HIP(hipStreamWaitEvent(dst.head, src.ev_head[(size_t)p], 0));        // packed on the owner
        GPX_HIP(hipMemcpyPeerAsync(dbuf, dst.dev, sbuf, src.dev, sizeof(double) * head_elems, dst.head));
        GPX_TRY(new_event(dst, dst.head, &dst.ev_head[(size_t)p]));
        src.readers[(size_t)p].push_back(dst.ev_head[(size_t)p]);                // one more reader of the OWNER's buffer
    }
    GPX_TRY(wait_slot_free(dst, p, r_far));
    GPX_HIP(hipStreamWaitEvent(r_far, src.ev_tail[(size_t)p], 0));
    GPX_HIP(hipMemcpyPeerAsync(dbuf + head_elems, dst.dev, sbuf + head_elems, src.dev, sizeof(double) * (all_elems - head_elems), r_far));
    GPX_TRY(new_event(dst, r_far, &dst.ev_tail[(size_t)p]));
    src.readers[(size_t)p].push_back(dst.ev_tail[(size_t)p]);
    if (hr == 0) dst.ev_head[(size_t)p] = dst.ev_tail[(size_t)p];
    const Views v = views(g, dst, p);
    dst.operand[(size_t)p].ptr = v.lower;
    dst.operand[(size_t)p].ld = q.w;
    dst.operand[(size_t)p].first = q.c0 + q.w;
    // the complete factor on every device: off the critical path, on the copy stream
    GPX_HIP(hipStreamWaitEvent(dst.copy, dst.ev_head[(size_t)p], 0));
    GPX_HIP(hipStreamWaitEvent(dst.copy, dst.ev_tail[(size_t)p], 0));
    if (below > 0)
        GPX_HIP(hipMemcpy2DAsync(dst.L + (q.c0 + q.w) * g->npad + q.c0, sizeof(double) * g->npad, v.lower, sizeof(double) * q.w, sizeof(double) * q.w, below,
                                 hipMemcpyDeviceToDevice, dst.copy));
    GPX_HIP(hipMemcpy2DAsync(dst.L + q.c0 * g->npad + q.c0, sizeof(double) * g->npad, v.square, sizeof(double) * q.w, sizeof(double) * q.w, q.w,
                             hipMemcpyDeviceToDevice, dst.copy));
    GPX_HIP(hipMemcpyAsync(dst.Dinv + q.b0 * (int64_t)TILE * TILE, v.dinv, sizeof(double) * (q.b1 - q.b0) * TILE * TILE, hipMemcpyDeviceToDevice, dst.copy));
    GPX_HIP(hipMemcpyAsync(dst.diag + q.c0, v.diag, sizeof(double) * q.w, hipMemcpyDeviceToDevice, dst.copy));
    hipEvent_t e;
    GPX_TRY(new_event(dst, dst.copy, &e));
    dst.readers[(size_t)p].push_back(e);
    return 0;
}

// apply panel p to the owned panels qs (ascending, all > p + 1) on the device's main stream
int update_panels(gpx_multi *g, MDev &m, const std::vector<int64_t> &qs, int64_t p)
{
    if (qs.empty()) return 0;
    GPX_TRY(set_dev(m));
    GPX_HIP(hipStreamWaitEvent(m.main, m.ev_tail[(size_t)p], 0));
    GPX_HIP(hipStreamWaitEvent(m.main, m.ev_head[(size_t)p], 0));
    const Operand &o = m.operand[(size_t)p];
    const Geom qp = geom(g, p);
    // consecutive owned panels form one launch (one device: all of them = the bulk SYRK of csrc/chol.hip)
    std::vector<std::pair<int64_t, int64_t>> runs;
    for (int64_t q : qs) {
        const Geom gq = geom(g, q);
        if (!runs.empty() && runs.back().second == gq.c0) runs.back().second = gq.c0 + gq.w;
        else runs.push_back({gq.c0, gq.c0 + gq.w});
    }
    for (size_t i = 0; i < runs.size(); ++i) {
        const int64_t q0 = runs[i].first, q1 = runs[i].second;
        const double *A = o.ptr + (q0 - o.first) * o.ld;
        GPX_TRY(gpx_dev_gemm_nt(A, o.ld, A, o.ld, m.L + q0 * g->npad + q0, g->npad, g->npad - q0, q1 - q0, qp.w, -1.0, 1.0, q1 == g->npad ? 1 : 0, m.main));
        // panel p + 2 now lacks panel p + 1 only: its owner's chain waits for THIS, not for the rest of the step's updates
        if (i == 0 && qs[0] == p + 2) GPX_TRY(new_event(m, m.main, &m.ev_look[(size_t)(p + 2)]));
    }
    hipEvent_t e;
    GPX_TRY(new_event(m, m.main, &e));
    m.readers[(size_t)p].push_back(e);
    return 0;
}

void release_device(MDev &m)
{
    (void)hipSetDevice(m.dev);
    (void)gpx_set_device(m.dev);
    if (m.h) { gpx_free(m.h); m.h = nullptr; }
    for (hipStream_t s : {m.main, m.side, m.copy, m.head, m.far})
        if (s) (void)hipStreamSynchronize(s);
    for (hipEvent_t e : m.all_events) (void)hipEventDestroy(e);
    m.all_events.clear();
    if (m.ev_built) { (void)hipEventDestroy(m.ev_built); m.ev_built = nullptr; }
    dfree(m.xw); dfree(m.L); dfree(m.Dinv); dfree(m.diag);
    for (int i = 0; i < NSLOTS; ++i) dfree(m.stage[i]);
    dfree(m.info);
    m.xw = m.L = m.Dinv = m.diag = nullptr;
    m.info = nullptr;
    for (int i = 0; i < NSLOTS; ++i) m.stage[i] = nullptr;
    if (m.main) stream_release(m.main, 0);
    if (m.side) stream_release(m.side, 1);
    if (m.copy) stream_release(m.copy, 0);
    if (m.head) stream_release(m.head, 1);
    if (m.far) stream_release(m.far, 1);
    m.main = m.side = m.copy = m.head = m.far = nullptr;
}

// one attempt of the sharded fit with the given jitter; *info_out = max of the devices' status words
int factor_all(gpx_multi *g, const double *x_host, int *info_out)
{
    const int R = (int)g->devs.size();
    const int64_t P = g->npanels;
    std::vector<double> xw((size_t)(g->n * g->d));
    for (int64_t i = 0; i < g->n; ++i)
        for (int k = 0; k < g->d; ++k) xw[(size_t)(i * g->d + k)] = x_host[i * g->d + k] * std::sqrt(std::exp(g->theta[2 + k]));
    for (MDev &m : g->devs) {
        GPX_TRY(set_dev(m));
        for (hipEvent_t e : m.all_events) (void)hipEventDestroy(e);   // (a retry: the first attempt's events)
        m.all_events.clear();
        m.ev_head.assign((size_t)P, nullptr);
        m.ev_tail.assign((size_t)P, nullptr);
        m.ev_look.assign((size_t)P + 2, nullptr);
        m.readers.assign((size_t)P, {});
        m.operand.assign((size_t)P, Operand());
        GPX_HIP(hipMemcpyAsync(m.xw, xw.data(), sizeof(double) * xw.size(), hipMemcpyHostToDevice, m.main));
        GPX_HIP(hipMemsetAsync(m.info, 0, 8 * sizeof(int), m.main));
    }
    for (MDev &m : g->devs) {   // xw is a host vector of this scope
        GPX_HIP(hipSetDevice(m.dev));
        GPX_HIP(hipStreamSynchronize(m.main));
    }
    for (int r = 0; r < R; ++r) {
        MDev &m = g->devs[(size_t)r];
        for (int64_t p = r; p < P; p += R) GPX_TRY(build_panel(g, m, p));
        GPX_TRY(set_dev(m));
        if (m.ev_built) (void)hipEventDestroy(m.ev_built);
        GPX_HIP(hipEventCreateWithFlags(&m.ev_built, hipEventDisableTiming));
        GPX_HIP(hipEventRecord(m.ev_built, m.main));
    }
    auto post = [&](int64_t p, int64_t prev) -> int {
        MDev &o = g->devs[(size_t)owner(g, p)];
        GPX_TRY(factor_panel(g, o, p, prev));
        for (int r = 0; r < R; ++r)
            if (r != owner(g, p)) GPX_TRY(receive_panel(g, o, g->devs[(size_t)r], p));
        return 0;
    };
    GPX_TRY(gpx_dev_set_panel_share(R));   // an owner's chain runs next to 1 / R of the trailing update (chol.hip, owner's step)
    GPX_TRY(post(0, -1));
    for (int64_t p = 0; p < P; ++p) {
        // host order: the (single, cheap to queue) trailing updates first, then the next panel's long chain of small launches -- on the
        // devices they run side by side on their own streams, ordered by events only
        for (int r = 0; r < R; ++r) {
            std::vector<int64_t> qs;
            for (int64_t q = r; q < P; q += R)
                if (q > p + 1) qs.push_back(q);
            GPX_TRY(update_panels(g, g->devs[(size_t)r], qs, p));
        }
        if (p + 1 < P) GPX_TRY(post(p + 1, p));
    }
    (void)gpx_dev_set_panel_share(1);
    int worst = 0;
    for (MDev &m : g->devs) {
        GPX_HIP(hipSetDevice(m.dev));
        for (hipStream_t s : {m.main, m.side, m.copy, m.head, m.far}) GPX_HIP(hipStreamSynchronize(s));
        int info = 0;
        GPX_HIP(hipMemcpy(&info, m.info, sizeof(int), hipMemcpyDeviceToHost));
        worst = std::max(worst, info);
    }
    *info_out = worst;
    return 0;
}

}   // namespace

// ---- C-ABI -------------------------------------------------------------------------------------------------------------------------------
// the calling thread's device choice (the library's thread-local one and HIP's) is put back when a multi-device call returns
struct DeviceGuard {
    int gpx_dev = 0, hip_dev = 0;
    DeviceGuard() : gpx_dev(gpx_thread_device()) { (void)hipGetDevice(&hip_dev); }
    ~DeviceGuard() { (void)gpx_set_device(gpx_dev); (void)hipSetDevice(hip_dev); }
};

extern "C" void gpx_multi_free(gpx_multi *g)
{
    if (!g) return;
    DeviceGuard guard;
    for (MDev &m : g->devs) release_device(m);
    delete g;
}

extern "C" int gpx_multi_fit(const double *x, const double *t_centered, int64_t n, int d, const double *theta, const int *devices, int ndev,
                             gpx_multi **out)
{
    if (out) *out = nullptr;
    if (!x || !t_centered || !theta || !devices || !out || n < 1 || d < 1 || d > GPX_MAX_D || ndev < 1 || ndev > 64) {
        gpx_set_error("gpx_multi_fit: bad arguments (n=%ld d=%d ndev=%d)", (long)n, d, ndev);
        return GPX_ERR_BAD_ARG;
    }
    for (int k = 0; k < 2 + d; ++k)
        if (std::isnan(theta[k]) || (k != 1 && !std::isfinite(theta[k]))) { gpx_set_error("gpx_multi_fit: theta[%d] is not finite", k); return GPX_ERR_BAD_ARG; }
    const int count = gpx_device_count();
    for (int r = 0; r < ndev; ++r)
        if (devices[r] < 0 || devices[r] >= count) {
            gpx_set_error("gpx_multi_fit: device %d not available (%d visible)", devices[r], count);
            return count == 0 ? GPX_ERR_NO_DEVICE : GPX_ERR_BAD_ARG;
        }
    gpx_multi *g = new gpx_multi();
    g->n = n; g->d = d;
    g->npad = (n + TILE - 1) / TILE * TILE;
    g->nblk = g->npad / TILE;
    g->npanels = (g->nblk + PB - 1) / PB;
    for (int k = 0; k < 2 + d; ++k) g->theta[k] = theta[k];
    g->v = std::exp(theta[0]);
    g->vt = std::exp(theta[1]);          // theta[1] = -inf: vt = 0
    g->devs.resize((size_t)ndev);
    int rc = 0;
    DeviceGuard guard;
    auto body = [&]() -> int {
        for (int r = 0; r < ndev; ++r) {
            MDev &m = g->devs[(size_t)r];
            m.dev = devices[r];
            GPX_TRY(set_dev(m));
            GPX_TRY(gpx_require_device());
            for (int q = 0; q < ndev; ++q)   // direct sends over xGMI ("already enabled" is fine; no peer path: the copy is staged by the runtime)
                if (devices[q] != m.dev) { (void)hipDeviceEnablePeerAccess(devices[q], 0); (void)hipGetLastError(); }
            // streams from the library's per-device cache (creating and destroying three streams costs more than a small fit); the chain's
            // stream in the high-priority class, like the single-GPU schedule's
            m.main = stream_acquire(0);
            m.side = stream_acquire(1);
            m.copy = stream_acquire(0);
            // (the two row streams only where the message is split: every further stream of a process changes which of them share a hardware
            // queue -- with five streams instead of three a ONE-rank fit at C3 took 42 ms instead of 29)
            if (split_message(g)) {
                m.head = stream_acquire(1);
                m.far = stream_acquire(1);
            }
            if (!m.main || !m.side || !m.copy || (split_message(g) && (!m.head || !m.far))) { gpx_set_error("gpx_multi_fit: no stream on device %d", m.dev); return GPX_ERR_HIP; }
            GPX_TRY(dalloc(&m.xw, n * d));
            GPX_TRY(dalloc(&m.L, g->npad * g->npad));
            GPX_TRY(dalloc(&m.Dinv, g->nblk * (int64_t)TILE * TILE));
            GPX_TRY(dalloc(&m.diag, g->npad));
            if (ndev > 1)
                for (int i = 0; i < NSLOTS; ++i) GPX_TRY(dalloc(&m.stage[i], message_elems(g, 0)));
            {
                double *w = nullptr;
                GPX_TRY(dalloc(&w, 8));
                m.info = reinterpret_cast<int *>(w);
            }
        }
        int info = 0;
        for (double jitter : {0.0, 1e-5}) {
            g->jitter = jitter;
            GPX_TRY(factor_all(g, x, &info));
            if (info == 0) break;
            if (info == GPX_INFO_STALLED) {
                gpx_set_error("gpx_multi_fit: a hand-off inside a panel step timed out (GPX_WAIT_LIMIT_MS); the factor is invalid");
                return GPX_ERR_STATE;
            }
        }
        if (info > 0) { gpx_set_error("covariance matrix not positive definite (leading minor %d), also with +1e-5 jitter", info); return info; }
        // every device holds the complete factor: a handle around it (alpha by the two-sweep solver) -- predict / propagate as usual
        for (MDev &m : g->devs) {
            GPX_TRY(set_dev(m));
            GPX_TRY(gpx_adopt_factor(x, t_centered, n, d, theta, m.L, m.Dinv, m.diag, g->jitter, nullptr, &m.h));
        }
        return 0;
    };
    rc = body();
    if (rc) { gpx_multi_free(g); return rc; }
    *out = g;
    return 0;
}

extern "C" int gpx_multi_info(const gpx_multi *g, int *ndev, int64_t *npanels, double *jitter)
{
    if (!g) { gpx_set_error("null handle"); return GPX_ERR_BAD_ARG; }
    if (ndev) *ndev = (int)g->devs.size();
    if (npanels) *npanels = g->npanels;
    if (jitter) *jitter = g->jitter;
    return 0;
}

extern "C" int gpx_multi_alpha(gpx_multi *g, double *beta_out)
{
    if (!g || !beta_out) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    DeviceGuard guard;
    GPX_TRY(gpx_set_device(g->devs[0].dev));
    return gpx_alpha(g->devs[0].h, beta_out);
}

// e3: the queries are dealt to the devices in contiguous shards, one host thread per device (gpx_predict is synchronous); mean WITHOUT meant
extern "C" int gpx_multi_predict(gpx_multi *g, const double *xs, int64_t m, double *mean_out, double *var_out)
{
    if (!g || !xs || !mean_out || !var_out || m < 0) { gpx_set_error("gpx_multi_predict: bad arguments"); return GPX_ERR_BAD_ARG; }
    if (m == 0) return 0;
    const int R = (int)g->devs.size();
    const int64_t per = (m + R - 1) / R;
    std::vector<int> rcs((size_t)R, 0);
    std::vector<std::string> errs((size_t)R);
    std::vector<std::thread> th;
    for (int r = 0; r < R; ++r) {
        const int64_t lo = std::min<int64_t>(m, r * per), hi = std::min<int64_t>(m, lo + per);
        if (hi <= lo) continue;
        th.emplace_back([&, r, lo, hi]() {
            int rc = gpx_set_device(g->devs[(size_t)r].dev);
            if (!rc) rc = gpx_predict(g->devs[(size_t)r].h, xs + lo * g->d, hi - lo, mean_out + lo, var_out + lo);
            rcs[(size_t)r] = rc;
            if (rc) errs[(size_t)r] = gpx_last_error();   // (thread-local text)
        });
    }
    for (std::thread &t : th) t.join();
    for (int r = 0; r < R; ++r)
        if (rcs[(size_t)r]) { gpx_set_error("gpx_multi_predict (device %d): %s", g->devs[(size_t)r].dev, errs[(size_t)r].c_str()); return rcs[(size_t)r]; }
    return 0;
}

// e4: UncertaintyPropagationApprox.propagate_GA (UncertaintyPropagation.py:397-479) with the d + 1 right-hand sides dealt to the devices;
// mean WITHOUT meant.  sigma2 / rest (optional) as gpx_propagate_approx returns them.
extern "C" int gpx_multi_propagate_approx(gpx_multi *g, const double *u, const double *Sigma, double *mean, double *var, double *sigma2, double *rest)
{
    if (!g || !u || !Sigma) { gpx_set_error("gpx_multi_propagate_approx: bad arguments"); return GPX_ERR_BAD_ARG; }
    const int R = (int)g->devs.size(), d = g->d, nvec = d + 1, npart = 4 + 2 * d;
    std::vector<std::vector<double>> part((size_t)R, std::vector<double>((size_t)npart, 0.0));
    std::vector<int> rcs((size_t)R, 0);
    std::vector<std::string> errs((size_t)R);
    std::vector<std::thread> th;
    const int per = nvec / R, extra = nvec % R;
    int k = 0;
    for (int r = 0; r < R; ++r) {
        const int cnt = per + (r < extra ? 1 : 0), k0 = k, k1 = k + cnt;
        k = k1;
        if (cnt == 0) continue;
        th.emplace_back([&, r, k0, k1]() {
            int rc = gpx_set_device(g->devs[(size_t)r].dev);
            if (!rc) rc = gpx_propagate_approx_rhs(g->devs[(size_t)r].h, u, Sigma, k0, k1, part[(size_t)r].data());
            rcs[(size_t)r] = rc;
            if (rc) errs[(size_t)r] = gpx_last_error();
        });
    }
    for (std::thread &t : th) t.join();
    for (int r = 0; r < R; ++r)
        if (rcs[(size_t)r]) { gpx_set_error("gpx_multi_propagate_approx (device %d): %s", g->devs[(size_t)r].dev, errs[(size_t)r].c_str()); return rcs[(size_t)r]; }
    std::vector<double> o((size_t)npart, 0.0);
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < npart; ++i) o[(size_t)i] += part[(size_t)r][(size_t)i];      // the one "all-reduce" of 4 + 2 d doubles
    // o = [beta.C, beta.tr, C.KinvC, KinvC.tr, (J_k.KinvJ_k, beta.J_k) for every k]   (UncertaintyPropagation.py:397-479)
    const double mu = o[0] + 0.5 * o[1];
    const double s2 = (g->v + g->vt) - o[2];
    double var2 = 0.0;
    for (int kk = 0; kk < d; ++kk) var2 -= Sigma[kk * d + kk] * (o[(size_t)(4 + 2 * kk)] - o[(size_t)(5 + 2 * kk)] * o[(size_t)(5 + 2 * kk)]);
    const double var3 = -o[3];
    if (mean) *mean = mu;
    if (var) *var = s2 + var2 + var3;
    if (sigma2) *sigma2 = s2;
    if (rest) *rest = var2 + var3;
    return 0;
}

// e4, Exact: UncertaintyPropagationExact.propagate_GA (UncertaintyPropagation.py:246-379) -- the j <= i double sum over
// (Kinv_ij - beta_i beta_j) L_ij cut into 128-aligned row panels of (almost) equal AREA of the triangle, one per device
// (gpx_propagate_exact_rows: a device builds only ITS rows of K^-1), two partial sums added on the host; mean WITHOUT meant.
extern "C" int gpx_multi_propagate_exact(gpx_multi *g, const double *u, const double *Sigma, double *mean, double *var)
{
    if (!g || !u || !Sigma) { gpx_set_error("gpx_multi_propagate_exact: bad arguments"); return GPX_ERR_BAD_ARG; }
    const int R = (int)g->devs.size();
    const int64_t nblk = (g->n + TILE - 1) / TILE;
    std::vector<int64_t> cuts((size_t)R + 1);
    for (int r = 0; r <= R; ++r) cuts[(size_t)r] = (int64_t)std::llround((double)nblk * std::sqrt((double)r / (double)R));
    cuts[0] = 0;
    cuts[(size_t)R] = nblk;
    std::vector<std::vector<double>> part((size_t)R, std::vector<double>(3, 0.0));
    std::vector<int> rcs((size_t)R, 0), used((size_t)R, 0);
    std::vector<std::string> errs((size_t)R);
    std::vector<std::thread> th;
    for (int r = 0; r < R; ++r) {
        const int64_t lo = std::min<int64_t>(g->n, cuts[(size_t)r] * TILE), hi = std::min<int64_t>(g->n, std::max(cuts[(size_t)r], cuts[(size_t)r + 1]) * TILE);
        if (hi <= lo) continue;
        used[(size_t)r] = 1;
        th.emplace_back([&, r, lo, hi]() {
            int rc = gpx_set_device(g->devs[(size_t)r].dev);
            if (!rc) rc = gpx_propagate_exact_rows(g->devs[(size_t)r].h, u, Sigma, lo, hi, part[(size_t)r].data());
            rcs[(size_t)r] = rc;
            if (rc) errs[(size_t)r] = gpx_last_error();
        });
    }
    for (std::thread &t : th) t.join();
    double p0 = 0.0, p1 = 0.0, nc2 = 0.0;
    for (int r = 0; r < R; ++r) {
        if (rcs[(size_t)r]) { gpx_set_error("gpx_multi_propagate_exact (device %d): %s", g->devs[(size_t)r].dev, errs[(size_t)r].c_str()); return rcs[(size_t)r]; }
        if (!used[(size_t)r]) continue;
        p0 += part[(size_t)r][0];
        p1 += part[(size_t)r][1];
        nc2 = part[(size_t)r][2];
    }
    if (mean) *mean = p0;
    if (var) *var = (g->v + g->vt) - nc2 * p1 - p0 * p0;      // UncertaintyPropagation.py:377
    return 0;
}
