// ry_vc.cpp -- the window call of libry355.so: `VoiceChanger.convert_from_acoustic_feature`
// (/root/reference/realtime_voice_conversion/yukarin_wrapper/voice_changer.py:24-42) on the device.  stage-1 on the effective frames ->
// combine_silent (scatter into the silent block) -> decode_spectrogram (mc2sp = exp(mc @ M)) + floor -> stage-2, as one chain per window:
// six ring slots with pinned staging (ry_vc_submit / ry_vc_wait), two window lanes over clones of the predictor pair (ry_vc_set_lanes),
// the silence gate on the device (ry_vc_gate / ry_vc_submit_wave), several windows per call (ry_vc_enqueue_device_batch), and the same
// chain cut where the reference's own class cuts it (ry_vc_stage1 / ry_vc_stage2_from_mc / ry_vc_mid_sp).  The predictors themselves
// are driven through the C ABI of ry_net.cpp (ry_ac_convert, ry_sr_convert_rows, ry_net_clone).
#include "ry_vc_kernels.h"
#include "ry_host.h"

// ---- device-resident VoiceChanger core -----------------------------------------------------------

// One window in flight = one ring slot: pinned host staging (so that hipMemcpyAsync really is asynchronous and H2D of window
// i + 1 / D2H of window i - 1 run under the kernels of window i), the device intermediates, and two events.
struct VcSlot {
    float *h_x = nullptr, *h_mc = nullptr, *h_sp = nullptr;   // pinned
    int* h_row = nullptr;                                       // pinned
    // device-side silence gate (ry_vc_submit_wave): the raw wave and the full feature block go up, the mask and the count come back
    float *h_wave = nullptr, *d_wave = nullptr, *h_feat = nullptr, *d_feat = nullptr, *d_pow = nullptr;
    unsigned char *h_mask = nullptr, *d_mask = nullptr;
    int *h_count = nullptr, *d_count = nullptr;
    int cap_wave = 0;
    // every slot owns its buffers and grows on its own: a longer window never touches a slot that still holds a window in flight
    Arena bufs;
    std::vector<void*> pinned;
    int cap_eff = 0, cap_frames = 0;
    void free_pinned() { for (void* q : pinned) rt::hfree(q); pinned.clear(); }
    void free_pinned_one(void* q) {
        for (size_t i = 0; i < pinned.size(); ++i)
            if (pinned[i] == q) { rt::hfree(q); pinned.erase(pinned.begin() + (long)i); return; }
    }
    bool gated = false;      // the window in the slot came through ry_vc_submit_wave
    float *d_x = nullptr, *d_y1 = nullptr, *d_mc = nullptr, *d_sp = nullptr, *d_out = nullptr;
    int* d_row = nullptr;
    rt::Event ev_mid;        // stage-1 stream: the spectrogram of this window is in d_sp (and mc in h_mc)
    rt::Event ev_done;       // stage-2 stream: everything of this window is done (sp in h_sp / the caller's device block)
    bool used = false;       // ev_done has been recorded at least once
    int ticket = -1;         // ticket of the window occupying the slot (submit .. wait), -1 = free
    int n_eff = 0, n_frames = 0;
    int k0 = 0, k1 = 0;      // rows of the spectrogram this window really computed (ry_vc_set_discard): the others come back as zeros
};

struct ry_vc {
    static const int RING = 16;           // ring slots that exist; `ring` of them are in use: six up to three lanes, else two per lane
    static const int MAX_LANES = 8;
    int ring = 6;                         // windows in flight (stage 1 of a lane's next window runs under stage 2 of its previous one)
    ry_net* s1 = nullptr;
    ry_net* s2 = nullptr;
    int M = 0, F = 0;
    Arena arena;
    float* d_mtx = nullptr;
    VcSlot slot[RING];
    bool has_ev = false;
    int next_ticket = 0;
    int dev_count = 0;       // device-pointer calls (ry_vc_enqueue_device) take the slots round robin
    int split_eff = -1;      // ry_vc_stage1 left the converted rows of this many effective frames in slot 0's d_y1 (-1: nothing)
    // Lanes (ry_vc_set_lanes): ring slot k runs on the predictor pair l1 / l2 [k % lanes].  Lane 0 is the caller's pair; the others are
    // clones (same filters, own streams / plans / activations), so that the windows in flight really run side by side: the tails of one
    // window's one-round grids and its weight-streaming bottom layers are filled by the other windows' kernels.
    int disc_front = 0, disc_back = 0;   // ry_vc_set_discard: frames of every window the caller throws away (stage 2 does not compute them)
    int lanes = 1;
    ry_net* l1[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ry_net* l2[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // a clone follows the arithmetic mode of the handle it was made from (ry_net_set_dtype on the caller's handle converts the filters
    // once; the clone takes the pointers and drops its launch plans)
    static ry_net* follow(ry_net* n, ry_net* src) {
        if (n != src && n->dtype != src->dtype) {
            rt::stream_sync(n->stream);
            n->layers = src->layers; n->dtype = src->dtype; n->plans.clear();
        }
        return n;
    }
    ry_net* lane1(int slot) const { return follow(l1[slot % lanes], s1); }
    ry_net* lane2(int slot) const { return follow(l2[slot % lanes], s2); }
    void sync_lanes() { for (int k = 0; k < lanes; ++k) { rt::stream_sync(l1[k]->stream); rt::stream_sync(l2[k]->stream); } }
    // several windows per call (ry_vc_enqueue_device_batch): their own intermediates, re-allocated when a larger batch arrives
    // ... TWO sets that take turns, so that stage 1 of one call runs under stage 2 of the call before it (with one set it had to wait
    // for that stage 2 to leave the buffers: every call paid its whole stage 1 in the open)
    struct BatchSet {
        Arena bufs;
        float *y1 = nullptr, *sp = nullptr;
        size_t cap_y1 = 0, cap_sp = 0;
        rt::Event mid, done;
        bool used = false;
    };
    BatchSet bset[2];
    int b_turn = 0;
    bool b_ev = false;
    void free_pinned() { for (VcSlot& sl : slot) sl.free_pinned(); }
};

static int vc_halloc(VcSlot& sl, void** p, size_t bytes) {
    rt::err_t e = rt::hmalloc(p, bytes);
    if (e != 0) return fail(RY_ENOMEM, "pinned host allocation of %zu bytes failed: %s", bytes, rt::err_str(e));
    sl.pinned.push_back(*p);
    return RY_OK;
}

// Size ONE ring slot for a window (grown on demand; the other slots, and the windows they may hold, are not touched).
static int vc_reserve_slot(ry_vc* vc, VcSlot& sl, int n_eff, int n_frames) {
    if (n_eff <= sl.cap_eff && n_frames <= sl.cap_frames) return RY_OK;
    if (sl.ticket >= 0) return fail(RY_ESTATE, "ring slot is held by ticket %d: ry_vc_wait it first", sl.ticket);
    if (sl.used) RT_TRY(rt::event_sync(sl.ev_done));           // a device-pointer window may still be running on the slot's buffers
    if (&sl == &vc->slot[0]) { rt::stream_sync(vc->s1->stream); rt::stream_sync(vc->s2->stream); }   // the split calls work on slot 0 without events
    sl.bufs.release(); sl.free_pinned();
    const int ce = n_eff > sl.cap_eff ? n_eff : sl.cap_eff, cf = n_frames > sl.cap_frames ? n_frames : sl.cap_frames;
    sl.cap_eff = sl.cap_frames = 0;
    const int cin = vc->s1->desc.in_ch;
    const size_t e1 = (size_t)(ce > 0 ? ce : 1);
    float* rowbuf = nullptr;
    RY_TRY(sl.bufs.alloc(&sl.d_x, e1 * cin));
    RY_TRY(sl.bufs.alloc(&sl.d_y1, e1 * vc->M));
    RY_TRY(sl.bufs.alloc(&rowbuf, e1));
    RY_TRY(sl.bufs.alloc(&sl.d_mc, (size_t)cf * vc->M));
    RY_TRY(sl.bufs.alloc(&sl.d_sp, (size_t)cf * vc->F));
    RY_TRY(sl.bufs.alloc(&sl.d_out, (size_t)cf * vc->F));
    sl.d_row = (int*)rowbuf;
    RY_TRY(vc_halloc(sl, (void**)&sl.h_x, e1 * cin * sizeof(float)));
    RY_TRY(vc_halloc(sl, (void**)&sl.h_row, e1 * sizeof(int)));
    RY_TRY(vc_halloc(sl, (void**)&sl.h_mc, (size_t)cf * vc->M * sizeof(float)));
    RY_TRY(vc_halloc(sl, (void**)&sl.h_sp, (size_t)cf * vc->F * sizeof(float)));
    // gate buffers: the feature block of ALL frames, one power per frame, the mask and the count (the wave buffer is sized on demand)
    float* q = nullptr;
    RY_TRY(sl.bufs.alloc(&sl.d_feat, (size_t)cf * cin));
    RY_TRY(sl.bufs.alloc(&q, (size_t)cf / 4 + 4)); sl.d_mask = (unsigned char*)q;
    RY_TRY(sl.bufs.alloc(&q, 4)); sl.d_count = (int*)q;
    RY_TRY(vc_halloc(sl, (void**)&sl.h_feat, (size_t)cf * cin * sizeof(float)));
    RY_TRY(vc_halloc(sl, (void**)&sl.h_mask, (size_t)cf + 16));
    RY_TRY(vc_halloc(sl, (void**)&sl.h_count, 16));
    sl.h_wave = nullptr; sl.d_wave = nullptr; sl.d_pow = nullptr; sl.cap_wave = 0;
    sl.used = false;
    sl.cap_eff = ce; sl.cap_frames = cf;
    if (&sl == &vc->slot[0]) vc->split_eff = -1;
    return RY_OK;
}

// every slot (ry_vc_reserve_frames: ahead of time, nothing may be in flight)
static int vc_reserve(ry_vc* vc, int n_eff, int n_frames) {
    for (int i = 0; i < vc->ring; ++i) { VcSlot& sl = vc->slot[i];
        if (sl.ticket >= 0 && (n_eff > sl.cap_eff || n_frames > sl.cap_frames))
            return fail(RY_ESTATE, "ticket %d is still in flight: ry_vc_wait it before reserving a larger ring", sl.ticket); }
    for (int i = 0; i < vc->ring; ++i) RY_TRY(vc_reserve_slot(vc, vc->slot[i], n_eff, n_frames));
    return RY_OK;
}

// scatter the converted rows into the all-silent block, then sp = exp(mc @ M) + floor   (stage-1 stream)
static int vc_enqueue_mid(ry_vc* vc, ry_net* s1, const float* y1, const int* row_of, int n_eff, int n_frames, float sp_floor, float* mc, float* sp) {
    ry_stream_t st1 = s1->stream;
    Launcher Lc{s1, s1->ctx, st1, nullptr, nullptr};
    const int M = vc->M, F = vc->F;
    RT_TRY(rt::dmemset(mc, 0, (size_t)n_frames * M * sizeof(float), st1));            // silent frames: zeros (AcousticFeature.silent)
    if (n_eff > 0) {
        RyScatterParams sc;
        sc.src = y1; sc.row_of = row_of; sc.dst = mc; sc.n_src = n_eff; sc.cols = M;
        dim3 sg((unsigned)(((long long)n_eff * M + 255) / 256));
        RY_TRY(Lc.begin("ry_scatter_rows", "combine_silent", 0, 0, sg));
        RY_LAUNCH(ry_scatter_rows, sg, 256, st1, sc);
        RY_TRY(Lc.end());
    }
    RyMc2spParams mp;
    mp.mc = mc; mp.mtx = vc->d_mtx; mp.sp = sp; mp.n = n_frames; mp.m = M; mp.f = F; mp.floor = sp_floor;
    dim3 mg((unsigned)(((long long)n_frames * F + 255) / 256));
    RY_TRY(Lc.begin("ry_mc2sp", "decode_spectrogram", 0, 0, mg));
    RY_LAUNCH(ry_mc2sp, mg, 256, st1, mp);
    RY_TRY(Lc.end());
    return RY_OK;
}

// stage 2 of one window on its lane
static int vc_run_stage2(ry_vc* vc, ry_net* s2, const float* sp_in, float* sp_out, int n_frames) {
    return ry_sr_convert_rows(s2, sp_in, sp_out, 1, n_frames, vc->disc_front, vc->disc_back, 1);
}

static int vc_check(const ry_vc* vc, const int* row_of, int n_eff, int n_frames, bool host_rows) {
    if (!vc) return fail(RY_EINVAL, "null argument");
    if (n_frames < 1 || n_eff < 0 || n_eff > n_frames) return fail(RY_EINVAL, "bad frame counts (%d effective of %d)", n_eff, n_frames);
    if (host_rows)
        for (int i = 0; i < n_eff; ++i)
            if (row_of[i] < 0 || row_of[i] >= n_frames) return fail(RY_EINVAL, "row_of[%d] = %d is outside the window", i, row_of[i]);
    return RY_OK;
}

// the rows of a window's spectrogram that are computed under the current ry_vc_set_discard (the same clipping as enqueue_forward)
static void vc_keep_rows(const ry_vc* vc, int n_frames, int* k0, int* k1) {
    *k0 = vc->disc_front < n_frames ? vc->disc_front : 0;
    *k1 = n_frames - vc->disc_back > *k0 ? n_frames - vc->disc_back : n_frames;
}

extern "C" {

int ry_vc_create(ry_net* s1, ry_net* s2, const float* mtx, int M, int F, ry_vc** out) {
    if (!s1 || !s2 || !mtx || !out) return fail(RY_EINVAL, "null argument");
    *out = nullptr;
    if (s1->desc.ndim != 1 || s2->desc.ndim != 2) return fail(RY_EINVAL, "ry_vc_create needs (stage-1, stage-2) predictors");
    if (s1->ctx != s2->ctx) return fail(RY_ESTATE, "both predictors must live in the same context");
    if (s1->desc.out_ch != M) return fail(RY_EINVAL, "stage-1 returns %d channels but the mc2sp matrix has %d rows", s1->desc.out_ch, M);
    if (s2->desc.width + 1 != F) return fail(RY_EINVAL, "stage-2 takes %d bins but the mc2sp matrix has %d columns", s2->desc.width + 1, F);
    RT_TRY(rt::set_device(s1->ctx->device));
    std::unique_ptr<ry_vc> vc(new ry_vc());
    vc->s1 = s1; vc->s2 = s2; vc->M = M; vc->F = F;
    vc->l1[0] = s1; vc->l2[0] = s2;
    std::vector<float> h(mtx, mtx + (size_t)M * F);
    RY_TRY(upload(vc->arena, s1->ctx, h, &vc->d_mtx));
    for (VcSlot& sl : vc->slot) { RT_TRY(rt::event_create_fast(&sl.ev_mid)); RT_TRY(rt::event_create_fast(&sl.ev_done)); }
    vc->has_ev = true;
    *out = vc.release();
    return RY_OK;
}

void ry_vc_destroy(ry_vc* vc) {
    if (!vc) return;
    rt::set_device(vc->s1->ctx->device);
    vc->sync_lanes();
    for (int k = 1; k < vc->lanes; ++k) { ry_net_destroy(vc->l1[k]); ry_net_destroy(vc->l2[k]); }
    if (vc->has_ev) for (VcSlot& sl : vc->slot) { rt::event_destroy(sl.ev_mid); rt::event_destroy(sl.ev_done); }
    if (vc->b_ev) for (ry_vc::BatchSet& bs : vc->bset) { rt::event_destroy(bs.mid); rt::event_destroy(bs.done); }
    vc->free_pinned();
    delete vc;
}

// The caller will throw away the first `front` and the last `back` frames of every window it gets back (ConvertStream.process does:
// it converts buffer + 2 x extra_time and picks the buffer, convert_stream.py:40-42).  Stage 2 then computes only the rows that are
// kept -- the decoder layers run on the row range those rows depend on, the encoder and the bottom of the U-Net stay whole -- and the
// discarded rows of the returned spectrogram are zero for the host-array calls (ry_vc_wait / ry_vc_wait_wave / ry_vc_stage2_from_mc); the
// device-pointer calls (ry_vc_enqueue_device / _batch) leave the discarded rows of the caller's block UNTOUCHED (no memset is queued).  The kept rows
// are bit-identical to the full result; mc is always complete.
// Applies to every following ry_vc_submit / ry_vc_submit_wave / ry_vc_enqueue_device / ry_vc_enqueue_device_batch until changed; (0, 0) =
// everything (ry_vc_stage2_from_mc included; ry_vc_mid_sp returns every row of the intermediate spectrogram).
int ry_vc_set_discard(ry_vc* vc, int front, int back) {
    if (!vc) return fail(RY_EINVAL, "null argument");
    if (front < 0 || back < 0 || front >= (1 << 20) || back >= (1 << 20)) return fail(RY_EINVAL, "bad discard counts (%d, %d)", front, back);
    vc->disc_front = front; vc->disc_back = back;
    return RY_OK;
}

// 1 .. 8 lanes (six ring slots up to three lanes, else two per lane; measured in round 3 at 300 frames: 1.133 / 1.166 / 1.156 / 1.147 / 1.121 ms per
// window with 2 / 3 / 4 / 6 / 8 lanes, profiles/r03/h_lane_experiments.txt; the default stays 2: the latency of a window grows with the lanes):
// ring slot k runs on its own pair of predictor handles (clones of the caller's: same filters, own streams, plans,
// activations and graphs), so that up to `lanes` windows really run side by side.  Measured at 300 frames (round 2):
// 1.281 / 1.200 / 1.160 ms per window with 1 / 2 / 3 lanes -- the one-round grids of one window leave tails and its bottom layers leave
// most of the chip idle; the other windows' kernels fill both.  Results do not change (same plans, same arithmetic).
int ry_vc_set_lanes(ry_vc* vc, int lanes) {
    if (!vc) return fail(RY_EINVAL, "null argument");
    if (lanes < 1 || lanes > ry_vc::MAX_LANES) return fail(RY_EINVAL, "lanes must be in 1..%d (got %d)", ry_vc::MAX_LANES, lanes);
    for (VcSlot& sl : vc->slot)
        if (sl.ticket >= 0) return fail(RY_ESTATE, "ticket %d is still in flight: ry_vc_wait it before changing the lanes", sl.ticket);
    RT_TRY(rt::set_device(vc->s1->ctx->device));
    vc->sync_lanes();
    while (vc->lanes > lanes) {
        --vc->lanes;
        ry_net_destroy(vc->l1[vc->lanes]); ry_net_destroy(vc->l2[vc->lanes]);
        vc->l1[vc->lanes] = vc->l2[vc->lanes] = nullptr;
    }
    while (vc->lanes < lanes) {
        ry_net *a = nullptr, *b = nullptr;
        RY_TRY(ry_net_clone(vc->s1, &a));
        int rc = ry_net_clone(vc->s2, &b);
        if (rc != RY_OK) { ry_net_destroy(a); return rc; }
        vc->l1[vc->lanes] = a; vc->l2[vc->lanes] = b;
        ++vc->lanes;
    }
    vc->ring = lanes <= 3 ? 6 : 2 * lanes;
    vc->dev_count = 0;
    return RY_OK;
}

// Host window in, ticket out: returns as soon as the copies and kernels are queued (nothing is waited for).
int ry_vc_submit(ry_vc* vc, const float* x_eff, const int* row_of, int n_eff, int n_frames, float sp_floor, int* ticket) {
    if (!vc || !ticket || (n_eff > 0 && (!x_eff || !row_of))) return fail(RY_EINVAL, "null argument");
    RY_TRY(vc_check(vc, row_of, n_eff, n_frames, true));
    RT_TRY(rt::set_device(vc->s1->ctx->device));
    const int t = vc->next_ticket;
    VcSlot& sl = vc->slot[t % vc->ring];
    ry_net *s1 = vc->lane1(t % vc->ring), *s2 = vc->lane2(t % vc->ring);
    if (sl.ticket >= 0) return fail(RY_ESTATE, "all %d ring slots are in flight: ry_vc_wait ticket %d first", vc->ring, sl.ticket);
    RY_TRY(vc_reserve_slot(vc, sl, n_eff, n_frames));
    const int cin = s1->desc.in_ch, M = vc->M, F = vc->F;
    ry_stream_t st1 = s1->stream, st2 = s2->stream;
    if (sl.used) RT_TRY(rt::stream_wait_event(st1, sl.ev_done));                        // (a device-pointer call may have used the slot last)
    if (n_eff > 0) {
        memcpy(sl.h_x, x_eff, (size_t)n_eff * cin * sizeof(float));
        memcpy(sl.h_row, row_of, (size_t)n_eff * sizeof(int));
        RT_TRY(rt::h2d(sl.d_x, sl.h_x, (size_t)n_eff * cin * sizeof(float), st1));
        RT_TRY(rt::h2d(sl.d_row, sl.h_row, (size_t)n_eff * sizeof(int), st1));
        RY_TRY(ry_ac_convert(s1, sl.d_x, sl.d_y1, 1, n_eff, 1));                       // stage-1 CNN on the effective frames
    }
    RY_TRY(vc_enqueue_mid(vc, s1, sl.d_y1, sl.d_row, n_eff, n_frames, sp_floor, sl.d_mc, sl.d_sp));
    RT_TRY(rt::d2h(sl.h_mc, sl.d_mc, (size_t)n_frames * M * sizeof(float), st1));
    RT_TRY(rt::event_record(sl.ev_mid, st1));
    RT_TRY(rt::stream_wait_event(st2, sl.ev_mid));                                      // stage-2 starts when the spectrogram is ready
    RY_TRY(vc_run_stage2(vc, s2, sl.d_sp, sl.d_out, n_frames));
    vc_keep_rows(vc, n_frames, &sl.k0, &sl.k1);
    RT_TRY(rt::d2h(sl.h_sp + (size_t)sl.k0 * F, sl.d_out + (size_t)sl.k0 * F, (size_t)(sl.k1 - sl.k0) * F * sizeof(float), st2));
    RT_TRY(rt::event_record(sl.ev_done, st2));
    sl.used = true; sl.ticket = t; sl.n_eff = n_eff; sl.n_frames = n_frames; sl.gated = false;
    vc->split_eff = -1;
    vc->next_ticket = t + 1;
    *ticket = t;
    return RY_OK;
}

int ry_vc_wait(ry_vc* vc, int ticket, float* mc_out, float* sp_out) {
    if (!vc || !mc_out || !sp_out) return fail(RY_EINVAL, "null argument");
    if (ticket < 0) return fail(RY_EINVAL, "bad ticket %d", ticket);
    VcSlot& sl = vc->slot[ticket % vc->ring];
    if (sl.ticket != ticket) return fail(RY_ESTATE, "ticket %d is not in flight (already waited for, or never submitted)", ticket);
    RT_TRY(rt::set_device(vc->s1->ctx->device));
    RT_TRY(rt::event_sync(sl.ev_done));            // ev_done follows ev_mid in stream order (stage-2 waited for it)
    memcpy(mc_out, sl.h_mc, (size_t)sl.n_frames * vc->M * sizeof(float));
    const size_t F = (size_t)vc->F;
    if (sl.k0 > 0) memset(sp_out, 0, (size_t)sl.k0 * F * sizeof(float));                       // frames the caller said it discards: not computed
    memcpy(sp_out + (size_t)sl.k0 * F, sl.h_sp + (size_t)sl.k0 * F, (size_t)(sl.k1 - sl.k0) * F * sizeof(float));
    if (sl.k1 < sl.n_frames) memset(sp_out + (size_t)sl.k1 * F, 0, (size_t)(sl.n_frames - sl.k1) * F * sizeof(float));
    sl.ticket = -1;
    return RY_OK;
}

int ry_vc_convert(ry_vc* vc, const float* x_eff, const int* row_of, int n_eff, int n_frames, float sp_floor,
                  float* mc_out, float* sp_out) {
    if (!vc || !mc_out || !sp_out) return fail(RY_EINVAL, "null argument");
    int t = -1;
    RY_TRY(ry_vc_submit(vc, x_eff, row_of, n_eff, n_frames, sp_floor, &t));
    return ry_vc_wait(vc, t, mc_out, sp_out);
}


// ---- the silence gate on the device: `separate_effective` + the gather of the effective rows, then the
// chain of ry_vc_submit.  The number of effective frames fixes the padded length of stage 1 (128 - n % 128), so the count is read
// back (4 bytes, pinned) before stage 1 is queued: one short wait per window in exchange for the host-side numpy gate.
}  // extern "C"

// wave + features up, frame powers, gate, compaction, count and mask back (waits for the count); leaves x_eff / row_of in the slot
static int vc_gate_into_slot(ry_vc* vc, ry_net* s1, VcSlot& sl, const float* wave, int n_samples, int hop, int fft_length, float p_effective, float p_all,
                             const float* feat, int n_frames, int* n_eff_out) {
    const int cin = s1->desc.in_ch;
    ry_stream_t st1 = s1->stream;
    if (sl.used) RT_TRY(rt::stream_wait_event(st1, sl.ev_done));
    const int n_wave_frames = n_samples / hop + 1;              // librosa: 1 + (len + 2 * (fft / 2) - fft) / hop
    if (n_samples > sl.cap_wave) {                              // wave staging of this slot, grown on demand: the previous buffers go
        RT_TRY(rt::stream_sync(st1));
        if (sl.d_wave) sl.bufs.free_one(sl.d_wave);
        if (sl.d_pow) sl.bufs.free_one(sl.d_pow);
        if (sl.h_wave) sl.free_pinned_one(sl.h_wave);
        sl.d_wave = sl.d_pow = sl.h_wave = nullptr; sl.cap_wave = 0;
        const int cap = n_samples + n_samples / 4 + 1024;
        RY_TRY(sl.bufs.alloc(&sl.d_wave, (size_t)cap));
        RY_TRY(sl.bufs.alloc(&sl.d_pow, (size_t)cap + 8));       // one power per WAVE frame (<= n_samples + 1 of them, whatever the hop)
        RY_TRY(vc_halloc(sl, (void**)&sl.h_wave, (size_t)cap * sizeof(float)));
        sl.cap_wave = cap;
    }
    memcpy(sl.h_wave, wave, (size_t)n_samples * sizeof(float));
    memcpy(sl.h_feat, feat, (size_t)n_frames * cin * sizeof(float));
    RT_TRY(rt::h2d(sl.d_wave, sl.h_wave, (size_t)n_samples * sizeof(float), st1));
    RT_TRY(rt::h2d(sl.d_feat, sl.h_feat, (size_t)n_frames * cin * sizeof(float), st1));
    Launcher Lc{s1, s1->ctx, st1, nullptr, nullptr};
    // The power of EVERY wave frame is taken: the host takes log_spec.max() (the top_db clamp) over all len(wave) // hop + 1 frames and
    // truncates to the feature block afterwards, so a frame past the block can still decide the clamp (a live window has n * hop samples:
    // always one frame more than features).  The mask and the compaction look at the first n_frames only (ry_gate_compact).
    RyFramePowerParams fp;
    fp.wave = sl.d_wave; fp.n = n_samples; fp.hop = hop; fp.fft = fft_length; fp.n_wave_frames = n_wave_frames; fp.power = sl.d_pow;
    dim3 fg((unsigned)((n_wave_frames + 3) / 4));
    RY_TRY(Lc.begin("ry_frame_power", "separate_effective", 0, 0, fg));
    RY_LAUNCH(ry_frame_power, fg, 256, st1, fp);
    RY_TRY(Lc.end());
    RyGateParams gp;
    gp.power = sl.d_pow; gp.n_wave_frames = n_wave_frames; gp.n_frames = n_frames; gp.p_eff = p_effective; gp.p_all = p_all;
    gp.feat = sl.d_feat; gp.cin = cin; gp.x_eff = sl.d_x; gp.row_of = sl.d_row; gp.count = sl.d_count; gp.mask = sl.d_mask;
    RY_TRY(Lc.begin("ry_gate_compact", "separate_effective", 0, 0, dim3(1)));
    RY_LAUNCH(ry_gate_compact, dim3(1), 1024, st1, gp);
    RY_TRY(Lc.end());
    RT_TRY(rt::d2h(sl.h_count, sl.d_count, sizeof(int), st1));
    RT_TRY(rt::d2h(sl.h_mask, sl.d_mask, (size_t)n_frames, st1));
    RT_TRY(rt::stream_sync(st1));                               // the count picks the stage-1 plan
    const int n_eff = sl.h_count[0];
    if (n_eff < 0 || n_eff > n_frames) return fail(RY_EHIP, "the gate returned %d effective frames of %d", n_eff, n_frames);
    *n_eff_out = n_eff;
    return RY_OK;
}

static int vc_gate_args(const ry_vc* vc, const float* wave, int n_samples, int hop, int fft_length, const float* feat, int n_frames) {
    if (!vc || !wave || !feat) return fail(RY_EINVAL, "null argument");
    if (n_samples < 1 || n_frames < 1 || hop < 1) return fail(RY_EINVAL, "bad sizes (%d samples, %d frames, hop %d)", n_samples, n_frames, hop);
    if (fft_length < 128 || fft_length > 1024 || (fft_length & (fft_length - 1)))
        return fail(RY_EINVAL, "the device gate takes fft_length 128 .. 1024, a power of two (got %d): use the host gate", fft_length);
    return RY_OK;
}

extern "C" {

// `AcousticConverter.separate_effective` alone: the mask, the count and (optionally) the gathered rows and their frame indices
int ry_vc_gate(ry_vc* vc, const float* wave, int n_samples, int hop, int fft_length, float p_effective, float p_all,
               const float* feat, int n_frames, unsigned char* effective_out, int* n_eff_out, float* x_eff_out, int* row_of_out) {
    RY_TRY(vc_gate_args(vc, wave, n_samples, hop, fft_length, feat, n_frames));
    if (!effective_out || !n_eff_out) return fail(RY_EINVAL, "null argument");
    RT_TRY(rt::set_device(vc->s1->ctx->device));
    RY_TRY(vc_reserve_slot(vc, vc->slot[0], n_frames, n_frames));
    VcSlot& sl = vc->slot[0];
    if (sl.ticket >= 0) return fail(RY_ESTATE, "ring slot 0 is held by ticket %d: ry_vc_wait it first", sl.ticket);
    int n_eff = 0;
    RY_TRY(vc_gate_into_slot(vc, vc->s1, sl, wave, n_samples, hop, fft_length, p_effective, p_all, feat, n_frames, &n_eff));
    memcpy(effective_out, sl.h_mask, (size_t)n_frames);
    *n_eff_out = n_eff;
    const int cin = vc->s1->desc.in_ch;
    if (n_eff > 0 && x_eff_out) RT_TRY(rt::d2h(x_eff_out, sl.d_x, (size_t)n_eff * cin * sizeof(float), vc->s1->stream));
    if (n_eff > 0 && row_of_out) RT_TRY(rt::d2h(row_of_out, sl.d_row, (size_t)n_eff * sizeof(int), vc->s1->stream));
    RT_TRY(rt::stream_sync(vc->s1->stream));
    vc->split_eff = -1;
    return RY_OK;
}

int ry_vc_submit_wave(ry_vc* vc, const float* wave, int n_samples, int hop, int fft_length, float p_effective, float p_all,
                      const float* feat, int n_frames, float sp_floor, int* ticket) {
    RY_TRY(vc_gate_args(vc, wave, n_samples, hop, fft_length, feat, n_frames));
    if (!ticket) return fail(RY_EINVAL, "null argument");
    RT_TRY(rt::set_device(vc->s1->ctx->device));
    const int t = vc->next_ticket;
    VcSlot& sl = vc->slot[t % vc->ring];
    ry_net *s1 = vc->lane1(t % vc->ring), *s2 = vc->lane2(t % vc->ring);
    if (sl.ticket >= 0) return fail(RY_ESTATE, "all %d ring slots are in flight: ry_vc_wait ticket %d first", vc->ring, sl.ticket);
    RY_TRY(vc_reserve_slot(vc, sl, n_frames, n_frames));
    const int M = vc->M, F = vc->F;
    ry_stream_t st1 = s1->stream, st2 = s2->stream;
    int n_eff = 0;
    RY_TRY(vc_gate_into_slot(vc, s1, sl, wave, n_samples, hop, fft_length, p_effective, p_all, feat, n_frames, &n_eff));
    if (n_eff > 0) RY_TRY(ry_ac_convert(s1, sl.d_x, sl.d_y1, 1, n_eff, 1));
    RY_TRY(vc_enqueue_mid(vc, s1, sl.d_y1, sl.d_row, n_eff, n_frames, sp_floor, sl.d_mc, sl.d_sp));
    RT_TRY(rt::d2h(sl.h_mc, sl.d_mc, (size_t)n_frames * M * sizeof(float), st1));
    RT_TRY(rt::event_record(sl.ev_mid, st1));
    RT_TRY(rt::stream_wait_event(st2, sl.ev_mid));
    RY_TRY(vc_run_stage2(vc, s2, sl.d_sp, sl.d_out, n_frames));
    vc_keep_rows(vc, n_frames, &sl.k0, &sl.k1);
    RT_TRY(rt::d2h(sl.h_sp + (size_t)sl.k0 * F, sl.d_out + (size_t)sl.k0 * F, (size_t)(sl.k1 - sl.k0) * F * sizeof(float), st2));
    RT_TRY(rt::event_record(sl.ev_done, st2));
    sl.used = true; sl.ticket = t; sl.n_eff = n_eff; sl.n_frames = n_frames; sl.gated = true;
    vc->split_eff = -1;
    vc->next_ticket = t + 1;
    *ticket = t;
    return RY_OK;
}

int ry_vc_wait_wave(ry_vc* vc, int ticket, float* mc_out, float* sp_out, unsigned char* effective_out, int* n_eff_out) {
    if (!vc || !effective_out) return fail(RY_EINVAL, "null argument");
    if (ticket < 0) return fail(RY_EINVAL, "bad ticket %d", ticket);
    VcSlot& sl = vc->slot[ticket % vc->ring];
    if (sl.ticket != ticket || !sl.gated) return fail(RY_ESTATE, "ticket %d is not a window submitted with ry_vc_submit_wave", ticket);
    memcpy(effective_out, sl.h_mask, (size_t)sl.n_frames);      // (already on the host: the submit waited for the count)
    if (n_eff_out) *n_eff_out = sl.n_eff;
    sl.gated = false;
    return ry_vc_wait(vc, ticket, mc_out, sp_out);
}

// Everything on the device, nothing waited for: consecutive calls pipeline by themselves (stage-1 of window i + 1 runs on
// its stream under stage-2 of window i); the intermediates rotate through the ring slots, ordered by events.
int ry_vc_enqueue_device(ry_vc* vc, const float* x_eff_dev, const int* row_of_dev, int n_eff, int n_frames, float sp_floor,
                         float* mc_out_dev, float* sp_out_dev) {
    if (!vc || !mc_out_dev || !sp_out_dev || (n_eff > 0 && (!x_eff_dev || !row_of_dev))) return fail(RY_EINVAL, "null argument");
    RY_TRY(vc_check(vc, nullptr, n_eff, n_frames, false));
    RT_TRY(rt::set_device(vc->s1->ctx->device));
    VcSlot& sl = vc->slot[vc->dev_count % vc->ring];
    ry_net *s1 = vc->lane1(vc->dev_count % vc->ring), *s2 = vc->lane2(vc->dev_count % vc->ring);
    if (sl.ticket >= 0) return fail(RY_ESTATE, "ring slot %d is held by ticket %d: ry_vc_wait it first", vc->dev_count % vc->ring, sl.ticket);
    RY_TRY(vc_reserve_slot(vc, sl, n_eff, n_frames));
    ++vc->dev_count;
    ry_stream_t st1 = s1->stream, st2 = s2->stream;
    if (sl.used) RT_TRY(rt::stream_wait_event(st1, sl.ev_done));                        // the slot's previous window has left d_sp
    if (n_eff > 0) RY_TRY(ry_ac_convert(s1, x_eff_dev, sl.d_y1, 1, n_eff, 1));
    RY_TRY(vc_enqueue_mid(vc, s1, sl.d_y1, row_of_dev, n_eff, n_frames, sp_floor, mc_out_dev, sl.d_sp));
    RT_TRY(rt::event_record(sl.ev_mid, st1));
    RT_TRY(rt::stream_wait_event(st2, sl.ev_mid));
    RY_TRY(vc_run_stage2(vc, s2, sl.d_sp, sp_out_dev, n_frames));
    RT_TRY(rt::event_record(sl.ev_done, st2));
    sl.used = true;
    vc->split_eff = -1;
    return RY_OK;
}

// Several independent windows of the same length in one call (streams served side by side, or the backlog of run.py's queue): stage 1
// runs as one batch when every window kept the same number of effective frames (else window by window), combine_silent per window,
// decode_spectrogram over all rows at once, stage 2 as ONE batch -- its bottom layers stream their filters once for all windows and
// every grid is many rounds of workgroups (measured at 300 frames: 1.28 ms for one window, 1.01 ms per window for eight).
// x_eff_dev / row_of_dev: the effective rows / row maps of the windows one after the other (sum of n_eff rows); n_eff: host array.
int ry_vc_enqueue_device_batch(ry_vc* vc, int n_windows, const float* x_eff_dev, const int* row_of_dev, const int* n_eff, int n_frames,
                               float sp_floor, float* mc_out_dev, float* sp_out_dev) {
    if (!vc || !n_eff || !mc_out_dev || !sp_out_dev) return fail(RY_EINVAL, "null argument");
    if (n_windows < 1 || n_windows > 4096) return fail(RY_EINVAL, "n_windows must be in 1..4096 (got %d)", n_windows);
    long long total_eff = 0;
    bool same = true;
    for (int w = 0; w < n_windows; ++w) {
        RY_TRY(vc_check(vc, nullptr, n_eff[w], n_frames, false));
        total_eff += n_eff[w];
        if (n_eff[w] != n_eff[0]) same = false;
    }
    if (total_eff > 0 && (!x_eff_dev || !row_of_dev)) return fail(RY_EINVAL, "null argument");
    ry_net *s1 = vc->s1, *s2 = vc->s2;
    RT_TRY(rt::set_device(s1->ctx->device));
    const int cin = s1->desc.in_ch, M = vc->M, F = vc->F;
    ry_stream_t st1 = s1->stream, st2 = s2->stream;
    if (!vc->b_ev) {
        for (ry_vc::BatchSet& q : vc->bset) { RT_TRY(rt::event_create_fast(&q.mid)); RT_TRY(rt::event_create_fast(&q.done)); }
        vc->b_ev = true;
    }
    const size_t need_y1 = (size_t)(total_eff > 0 ? total_eff : 1) * M, need_sp = (size_t)n_windows * n_frames * F;
    ry_vc::BatchSet& bs = vc->bset[vc->b_turn++ & 1];
    if (need_y1 > bs.cap_y1 || need_sp > bs.cap_sp) {
        RT_TRY(rt::stream_sync(st1)); RT_TRY(rt::stream_sync(st2));       // queued work may still use the old buffers
        bs.bufs.release();
        bs.cap_y1 = bs.cap_sp = 0; bs.used = false;
        RY_TRY(bs.bufs.alloc(&bs.y1, need_y1));
        RY_TRY(bs.bufs.alloc(&bs.sp, need_sp));
        bs.cap_y1 = need_y1; bs.cap_sp = need_sp;
    }
    if (bs.used) RT_TRY(rt::stream_wait_event(st1, bs.done));               // the call before the previous one has left this set
    if (total_eff > 0) {
        if (same) RY_TRY(ry_ac_convert(s1, x_eff_dev, bs.y1, n_windows, n_eff[0], 1));
        else {
            long long off = 0;
            for (int w = 0; w < n_windows; ++w) {
                if (n_eff[w] > 0) RY_TRY(ry_ac_convert(s1, x_eff_dev + off * cin, bs.y1 + off * M, 1, n_eff[w], 1));
                off += n_eff[w];
            }
        }
    }
    {   // combine_silent per window (its own row map), then decode_spectrogram over the rows of all windows in one launch
        Launcher Lc{s1, s1->ctx, st1, nullptr, nullptr};
        RT_TRY(rt::dmemset(mc_out_dev, 0, (size_t)n_windows * n_frames * M * sizeof(float), st1));
        long long off = 0;
        for (int w = 0; w < n_windows; ++w) {
            if (n_eff[w] > 0) {
                RyScatterParams sc;
                sc.src = bs.y1 + off * M; sc.row_of = row_of_dev + off; sc.dst = mc_out_dev + (size_t)w * n_frames * M; sc.n_src = n_eff[w]; sc.cols = M;
                dim3 sg((unsigned)(((long long)n_eff[w] * M + 255) / 256));
                RY_TRY(Lc.begin("ry_scatter_rows", "combine_silent", 0, 0, sg));
                RY_LAUNCH(ry_scatter_rows, sg, 256, st1, sc);
                RY_TRY(Lc.end());
            }
            off += n_eff[w];
        }
        RyMc2spParams mp;
        mp.mc = mc_out_dev; mp.mtx = vc->d_mtx; mp.sp = bs.sp; mp.n = n_windows * n_frames; mp.m = M; mp.f = F; mp.floor = sp_floor;
        dim3 mg((unsigned)(((long long)n_windows * n_frames * F + 255) / 256));
        RY_TRY(Lc.begin("ry_mc2sp", "decode_spectrogram", 0, 0, mg));
        RY_LAUNCH(ry_mc2sp, mg, 256, st1, mp);
        RY_TRY(Lc.end());
    }
    RT_TRY(rt::event_record(bs.mid, st1));
    RT_TRY(rt::stream_wait_event(st2, bs.mid));
    RY_TRY(ry_sr_convert_rows(s2, bs.sp, sp_out_dev, n_windows, n_frames, vc->disc_front, vc->disc_back, 1));
    RT_TRY(rt::event_record(bs.done, st2));
    bs.used = true;
    vc->split_eff = -1;
    return RY_OK;
}

// ---- the same chain cut where the reference's own VoiceChanger cuts it (voice_changer.py:33-41), so that its unchanged
// step-by-step calls still keep the data on the device between the two CNNs:
//   ry_vc_stage1          = AcousticConverter.convert            : H2D of the effective frames, stage-1, D2H of the converted rows
//   ry_vc_stage2_from_mc  = combine_silent + decode_spectrogram + `+ floor` + SuperResolution.convert, from the rows stage 1 left
//                           on the device: one D2H of the spectrogram; the intermediate spectrogram never visits the host
//   ry_vc_mid_sp          = the intermediate spectrogram, for a caller that does read it
int ry_vc_stage1(ry_vc* vc, const float* x_eff, int n_eff, float* y1_out) {
    if (!vc || !x_eff || !y1_out || n_eff < 1) return fail(RY_EINVAL, "bad argument");
    ry_net* s1 = vc->s1;
    RT_TRY(rt::set_device(s1->ctx->device));
    RY_TRY(vc_reserve_slot(vc, vc->slot[0], n_eff, n_eff));
    VcSlot& sl = vc->slot[0];
    if (sl.ticket >= 0) return fail(RY_ESTATE, "ring slot 0 is held by ticket %d: ry_vc_wait it first", sl.ticket);
    const int cin = s1->desc.in_ch;
    if (sl.used) RT_TRY(rt::stream_wait_event(s1->stream, sl.ev_done));
    memcpy(sl.h_x, x_eff, (size_t)n_eff * cin * sizeof(float));
    RT_TRY(rt::h2d(sl.d_x, sl.h_x, (size_t)n_eff * cin * sizeof(float), s1->stream));
    RY_TRY(ry_ac_convert(s1, sl.d_x, sl.d_y1, 1, n_eff, 1));
    RT_TRY(rt::d2h(sl.h_mc, sl.d_y1, (size_t)n_eff * vc->M * sizeof(float), s1->stream));
    RT_TRY(rt::stream_sync(s1->stream));
    memcpy(y1_out, sl.h_mc, (size_t)n_eff * vc->M * sizeof(float));
    vc->split_eff = n_eff;
    return RY_OK;
}

// the window is longer than the ring was sized for: grow it without losing the rows stage 1 left in slot 0
static int vc_grow_keep_rows(ry_vc* vc, int n_frames) {
    if (n_frames <= vc->slot[0].cap_frames) return RY_OK;
    const int keep = vc->split_eff;
    if (keep <= 0) return vc_reserve_slot(vc, vc->slot[0], vc->slot[0].cap_eff, n_frames);
    Arena tmp;
    float* t = nullptr;
    RY_TRY(tmp.alloc(&t, (size_t)keep * vc->M));
    RT_TRY(rt::d2d(t, vc->slot[0].d_y1, (size_t)keep * vc->M * sizeof(float), vc->s1->stream));
    RT_TRY(rt::stream_sync(vc->s1->stream));
    RY_TRY(vc_reserve_slot(vc, vc->slot[0], vc->slot[0].cap_eff, n_frames));
    RT_TRY(rt::d2d(vc->slot[0].d_y1, t, (size_t)keep * vc->M * sizeof(float), vc->s1->stream));
    RT_TRY(rt::stream_sync(vc->s1->stream));
    vc->split_eff = keep;
    return RY_OK;
}

static int vc_split_mid(ry_vc* vc, const int* row_of, int n_eff, int n_frames, float sp_floor) {
    RY_TRY(vc_check(vc, row_of, n_eff, n_frames, true));
    if (n_eff > 0 && !row_of) return fail(RY_EINVAL, "null argument");
    if (n_eff > 0 && vc->split_eff != n_eff)
        return fail(RY_ESTATE, "ry_vc_stage1 left %d converted rows on the device, this call asks for %d", vc->split_eff, n_eff);
    RT_TRY(rt::set_device(vc->s1->ctx->device));
    RY_TRY(vc_grow_keep_rows(vc, n_frames));
    VcSlot& sl = vc->slot[0];
    if (n_eff > 0) {
        memcpy(sl.h_row, row_of, (size_t)n_eff * sizeof(int));
        RT_TRY(rt::h2d(sl.d_row, sl.h_row, (size_t)n_eff * sizeof(int), vc->s1->stream));
    } else if (sl.used) {
        RT_TRY(rt::stream_wait_event(vc->s1->stream, sl.ev_done));
    }
    return vc_enqueue_mid(vc, vc->s1, sl.d_y1, sl.d_row, n_eff, n_frames, sp_floor, sl.d_mc, sl.d_sp);
}

int ry_vc_stage2_from_mc(ry_vc* vc, const int* row_of, int n_eff, int n_frames, float sp_floor, float* sp_out) {
    if (!vc || !sp_out) return fail(RY_EINVAL, "null argument");
    RY_TRY(vc_split_mid(vc, row_of, n_eff, n_frames, sp_floor));
    const int keep = vc->split_eff;
    VcSlot& sl = vc->slot[0];
    ry_net* s2 = vc->s2;
    RT_TRY(rt::event_record(sl.ev_mid, vc->s1->stream));
    RT_TRY(rt::stream_wait_event(s2->stream, sl.ev_mid));
    int k0 = 0, k1 = n_frames;
    vc_keep_rows(vc, n_frames, &k0, &k1);                       // ry_vc_set_discard: the rows the caller throws away are not computed
    const size_t F = (size_t)vc->F;
    RY_TRY(ry_sr_convert_rows(s2, sl.d_sp, sl.d_out, 1, n_frames, vc->disc_front, vc->disc_back, 1));
    RT_TRY(rt::d2h(sl.h_sp + (size_t)k0 * F, sl.d_out + (size_t)k0 * F, (size_t)(k1 - k0) * F * sizeof(float), s2->stream));
    RT_TRY(rt::event_record(sl.ev_done, s2->stream));
    sl.used = true;
    RT_TRY(rt::event_sync(sl.ev_done));
    if (k0 > 0) memset(sp_out, 0, (size_t)k0 * F * sizeof(float));
    memcpy(sp_out + (size_t)k0 * F, sl.h_sp + (size_t)k0 * F, (size_t)(k1 - k0) * F * sizeof(float));
    if (k1 < n_frames) memset(sp_out + (size_t)k1 * F, 0, (size_t)(n_frames - k1) * F * sizeof(float));
    vc->split_eff = keep;                                     // the rows stay valid until the next stage-1 call
    return RY_OK;
}

int ry_vc_mid_sp(ry_vc* vc, const int* row_of, int n_eff, int n_frames, float sp_floor, float* sp_mid_out) {
    if (!vc || !sp_mid_out) return fail(RY_EINVAL, "null argument");
    RY_TRY(vc_split_mid(vc, row_of, n_eff, n_frames, sp_floor));
    const int keep = vc->split_eff;
    VcSlot& sl = vc->slot[0];
    RT_TRY(rt::d2h(sl.h_sp, sl.d_sp, (size_t)n_frames * vc->F * sizeof(float), vc->s1->stream));
    RT_TRY(rt::stream_sync(vc->s1->stream));
    memcpy(sp_mid_out, sl.h_sp, (size_t)n_frames * vc->F * sizeof(float));
    vc->split_eff = keep;
    return RY_OK;
}

// Reserve the ring for windows of up to n_frames (all effective) ahead of time (optional: every call grows it on demand, the
// split calls at the price of a copy of the rows stage 1 left on the device).
int ry_vc_reserve_frames(ry_vc* vc, int n_frames) {
    if (!vc || n_frames < 1) return fail(RY_EINVAL, "bad argument");
    RT_TRY(rt::set_device(vc->s1->ctx->device));
    return vc_reserve(vc, n_frames, n_frames);
}

int ry_mc2sp(ry_ctx* ctx, const float* mc, const float* mtx, int n, int m, int bins, float floor_, float* sp) {
    if (!ctx || !mc || !mtx || !sp || n < 1 || m < 1 || bins < 1) return fail(RY_EINVAL, "bad argument");
    RT_TRY(rt::set_device(ctx->device));
    Arena a;
    float *dmc = nullptr, *dmtx = nullptr, *dsp = nullptr;
    RY_TRY(a.alloc(&dmc, (size_t)n * m)); RY_TRY(a.alloc(&dmtx, (size_t)m * bins)); RY_TRY(a.alloc(&dsp, (size_t)n * bins));
    RT_TRY(rt::h2d(dmc, mc, (size_t)n * m * sizeof(float), ctx->stream));
    RT_TRY(rt::h2d(dmtx, mtx, (size_t)m * bins * sizeof(float), ctx->stream));
    RyMc2spParams mp;
    mp.mc = dmc; mp.mtx = dmtx; mp.sp = dsp; mp.n = n; mp.m = m; mp.f = bins; mp.floor = floor_;
    dim3 mg((unsigned)(((long long)n * bins + 255) / 256));
    Launcher Lc{nullptr, ctx, ctx->stream, nullptr, nullptr};
    RY_TRY(Lc.begin("ry_mc2sp", "decode_spectrogram", 0, 0, mg));
    RY_LAUNCH(ry_mc2sp, mg, 256, ctx->stream, mp);
    RY_TRY(Lc.end());
    RT_TRY(rt::d2h(sp, dsp, (size_t)n * bins * sizeof(float), ctx->stream));
    RT_TRY(rt::stream_sync(ctx->stream));
    return RY_OK;
}

}  // extern "C"
