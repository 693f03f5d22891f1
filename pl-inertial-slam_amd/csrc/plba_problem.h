// plba_problem.h — host-side problem object behind the opaque plba_problem* of include/plba.h.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "plba_internal.h"

namespace plba {

// Device buffers come from a process-wide pool of power-of-two size classes (per device): a BA call allocates ~100
// buffers and the window changes with every call, and hipMalloc / hipFree (which synchronises the device) cost more
// wall time per call than the fifteen LM iterations.  Blocks go back to the pool on release and are only returned to
// the driver beyond POOL_CAP cached bytes.  The owner synchronises its stream before releasing (plba_destroy does).
struct DevPool {
    static constexpr size_t POOL_CAP = (size_t)8 << 30;
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void*>> free_;     // (device, size class) -> blocks
    size_t cached = 0;
    size_t n_malloc = 0, n_reuse = 0;      // statistics (options.diag & PLBA_DIAG_TIMING prints them)
    static size_t size_class(size_t bytes) { size_t c = 4096; while (c < bytes) c <<= 1; return c; }
    hipError_t get(size_t bytes, void** out, size_t* cls) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        *cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = free_.find({dev, *cls});
            if (it != free_.end() && !it->second.empty()) { *out = it->second.back(); it->second.pop_back(); cached -= *cls; ++n_reuse; return hipSuccess; }
            ++n_malloc;
        }
        hipError_t e = hipMalloc(out, *cls);
        if (e != hipSuccess) {      // give the cache back to the driver and try once more
            trim();
            e = hipMalloc(out, *cls);
        }
        return e;
    }
    void put(void* ptr, size_t cls) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(mu);
            if (cached + cls <= POOL_CAP) { free_[{dev, cls}].push_back(ptr); cached += cls; return; }
        }
        (void)hipFree(ptr);
    }
    void trim() {
        std::lock_guard<std::mutex> g(mu);
        for (auto& kv : free_) for (void* q : kv.second) (void)hipFree(q);
        free_.clear(); cached = 0;
    }
};
inline DevPool& dev_pool() { static DevPool* pool = new DevPool; return *pool; }      // never destroyed: no HIP calls at process exit
// stream the zero-fill of fresh buffers is ordered on (set by the API entry point for its duration); null: legacy
// synchronous hipMemset + drain of the null stream
inline hipStream_t& darr_stream() { static thread_local hipStream_t s = nullptr; return s; }
// pinned staging area the queued uploads go through (host vector -> staging by memcpy, staging -> device by an async
// copy): pageable sources make the runtime pin / stage on its own, which was measured at ~20 ms per BA call as soon as a
// second problem is alive in the process.  Bump-allocated; valid until the owner has synchronised the stream.
// [0, cap - XFER): bump-allocated by the queued uploads of one DArrStreamScope; [cap - XFER, cap): bounce buffer of the blocking
// plba_d2h / plba_h2d copies, which may run while queued uploads are still waiting for their turn on the stream
struct StageArea {
    char* base = nullptr; size_t cap = 0, used = 0;
    static constexpr size_t XFER = (size_t)4 << 20;
    size_t upload_cap() const { return cap > XFER ? cap - XFER : 0; }
    char* xfer() const { return base + upload_cap(); }
    size_t xfer_cap() const { return cap - upload_cap(); }
};
inline StageArea*& darr_stage() { static thread_local StageArea* a = nullptr; return a; }
struct DArrStreamScope {
    hipStream_t prev;
    StageArea* prev_stage;
    explicit DArrStreamScope(hipStream_t s, StageArea* st = nullptr) : prev(darr_stream()), prev_stage(darr_stage()) { darr_stream() = s; darr_stage() = st; if (st) st->used = 0; }
    ~DArrStreamScope() { darr_stream() = prev; darr_stage() = prev_stage; }
};

// a block of the current staging area to build an upload in place (null: no staging area, or full)
inline void* stage_take(size_t bytes) {
    StageArea* st = darr_stage();
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (!darr_stream() || !st || !st->base || st->used + need > st->upload_cap()) return nullptr;
    void* r = st->base + st->used;
    st->used += need;
    return r;
}

// prepare() creates ~150 buffers per BA call, most of them a few kilobytes (index maps, per-keyframe tables, the lists of the reduced
// system): one hipMemsetAsync or hipMemcpyAsync each is ~4 us of host time, 0.5 ms per call — a third of a 15-iteration optimize().
// Inside a DevBatchScope the SMALL ones are bump-allocated from two device blocks the problem keeps: `z` for the zero-filled ones
// (cleared by ONE hipMemsetAsync per flush) and `u` for the uploads (filled through a pinned mirror `uh` at the same offsets and copied by ONE
// hipMemcpyAsync per flush).  darr_flush() runs before anything on the stream reads them (every kernel launch inside prepare(), and its
// end); the blocks start over with the next prepare(), which re-creates every buffer that lives in them.
struct DevBatch {
    static constexpr size_t SMALL = (size_t)256 << 10, ZCAP = (size_t)8 << 20, UCAP = (size_t)4 << 20;
    char* z = nullptr; size_t zcap = 0, zused = 0, zdone = 0;
    char* u = nullptr; char* uh = nullptr; size_t ucap = 0, uused = 0, udone = 0;
    size_t n_batched = 0;
    unsigned gen = 0;      // bumped when the blocks start over (prepare()): a DArr stamped with an older generation points at memory that now belongs to other buffers
    static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
    void* take_zero(size_t bytes) { const size_t need = al(bytes); if (!z || zused + need > zcap) return nullptr; void* r = z + zused; zused += need; ++n_batched; return r; }
    bool take_upload(size_t bytes, void** dev, void** host) { const size_t need = al(bytes); if (!u || !uh || uused + need > ucap) return false; *dev = u + uused; *host = uh + uused; uused += need; ++n_batched; return true; }
};
inline DevBatch*& darr_batch() { static thread_local DevBatch* b = nullptr; return b; }
// where the zero-filled (non-batched) buffers of the current scope are noted, so that a window which re-uses the previous window's pose
// structure (prepare(): structure cache) can clear them again without allocating them again
inline std::vector<std::pair<void*, size_t>>*& darr_zero_log() { static thread_local std::vector<std::pair<void*, size_t>>* l = nullptr; return l; }
inline hipError_t darr_flush() {
    DevBatch* b = darr_batch();
    if (!b) return hipSuccess;
    hipError_t e = hipSuccess;
    if (b->zused > b->zdone) { e = hipMemsetAsync(b->z + b->zdone, 0, b->zused - b->zdone, darr_stream()); b->zdone = b->zused; }
    if (e == hipSuccess && b->uused > b->udone) { e = hipMemcpyAsync(b->u + b->udone, b->uh + b->udone, b->uused - b->udone, hipMemcpyHostToDevice, darr_stream()); b->udone = b->uused; }
    return e;
}

template <class T>
struct DArr {
    T* p = nullptr;
    size_t n = 0;
    size_t cls = 0;          // pool size class of the block behind p (batched: its aligned size)
    bool batched = false;    // p points into the problem's batch blocks (DevBatch), not at a pool block
    unsigned bgen = 0;       // ... of this generation of the blocks (a batched buffer that the next prepare() does not re-create must not stay usable)
    const DevBatch* bsrc = nullptr;      // ... of this batch (the per-window one or the pose structure's)
    void drop_batched() { if (batched) { p = nullptr; n = 0; cls = 0; batched = false; } }
    // null unless the buffer is a pool block or a batched one of the CURRENT generation: for the conditionally allocated buffers
    // (d_imu_loc, d_ob_err, d_Ninvd, d_pr_H, d_lmg_*), whose previous window's pointer would alias another buffer's memory
    T* checked() const { return (batched && bsrc && bgen != bsrc->gen) ? nullptr : p; }
    void expire() { if (batched && bsrc && bgen != bsrc->gen) drop_batched(); }
    hipError_t alloc(size_t cnt, bool zero = true) {
        if (cnt == 0) cnt = 1;
        if (DevBatch* b = darr_batch()) {
            drop_batched();      // the batch blocks start over with every prepare()
            if (zero && cnt * sizeof(T) <= DevBatch::SMALL)
                if (void* q = b->take_zero(cnt * sizeof(T))) { release(); p = (T*)q; n = cnt; cls = DevBatch::al(cnt * sizeof(T)); batched = true; bgen = b->gen; bsrc = b; return hipSuccess; }
        }
        if (!p || cnt * sizeof(T) > cls) {
            release();
            hipError_t e = dev_pool().get(cnt * sizeof(T), (void**)&p, &cls);
            if (e != hipSuccess) { p = nullptr; cls = 0; return e; }
        }
        n = cnt;
        if (zero) {
            if (auto* zl = darr_zero_log()) zl->push_back({(void*)p, n * sizeof(T)});
            if (hipStream_t s = darr_stream()) return hipMemsetAsync(p, 0, n * sizeof(T), s);
            // hipMemset runs on the legacy null stream and may return before it completes; the library's
            // kernels run on a NON-blocking stream that does not order against it, so drain it here.
            hipError_t e = hipMemset(p, 0, n * sizeof(T));
            if (e != hipSuccess) return e;
            return hipStreamSynchronize(nullptr);
        }
        return hipSuccess;
    }
    // With a stream set (DArrStreamScope) the copy is queued on it: the CALLER keeps `h` alive and unchanged until it
    // has synchronised that stream (prepare() does, once, at its end) instead of paying a synchronisation per buffer.
    hipError_t upload(const std::vector<T>& h) {
        if (DevBatch* b = darr_batch()) {
            void *dev, *host;
            if (!h.empty() && h.size() * sizeof(T) <= DevBatch::SMALL && b->take_upload(h.size() * sizeof(T), &dev, &host)) {
                drop_batched(); release();
                p = (T*)dev; n = h.size(); cls = DevBatch::al(h.size() * sizeof(T)); batched = true; bgen = b->gen; bsrc = b;
                memcpy(host, h.data(), h.size() * sizeof(T));
                return hipSuccess;
            }
        }
        hipError_t e = alloc(h.size(), h.empty());
        if (e != hipSuccess || h.empty()) return e;
        const size_t bytes = h.size() * sizeof(T);
        if (hipStream_t s = darr_stream()) {
            StageArea* st = darr_stage();
            const size_t need = (bytes + 255) & ~(size_t)255;
            if (st && st->base && st->used + need <= st->upload_cap()) {
                char* src = st->base + st->used;
                st->used += need;
                memcpy(src, h.data(), bytes);
                return hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s);
            }
            return hipMemcpyAsync(p, h.data(), bytes, hipMemcpyHostToDevice, s);
        }
        return hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice);
    }
    // `src` already lives in the pinned staging area (stage_take below): one asynchronous copy, no host-side memcpy
    hipError_t upload_staged(const T* src, size_t cnt) {
        hipError_t e = alloc(cnt, cnt == 0);
        if (e != hipSuccess || cnt == 0) return e;
        return hipMemcpyAsync(p, src, cnt * sizeof(T), hipMemcpyHostToDevice, darr_stream());
    }
    void release() { if (p && !batched) dev_pool().put(p, cls); p = nullptr; n = 0; cls = 0; batched = false; }
    void swap(DArr& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cls, o.cls); std::swap(batched, o.batched); std::swap(bgen, o.bgen); std::swap(bsrc, o.bsrc); }
    DArr() = default;
    DArr(const DArr&) = delete;
    DArr& operator=(const DArr&) = delete;
    ~DArr() { release(); }
};

}  // namespace plba

namespace plba {
// host image of the fused landmark-major passes' group structure (build_lm_groups, plba_api.hip); kept with the cached context
// (HostCtx) so that its tables are not re-allocated (and their pages not re-faulted) by every BA call
struct LmHost {
    std::vector<LmGroup> grp;
    int wmax = 8, npair = 36;      // window slots / pose-pair blocks per group in the gather buffer: 16 / 136 when wide groups exist
    std::vector<int32_t> lm_slot, lm_ob0, ob_orig, blk_ij, blk_start, blk_src, row_kf, row_start, row_src;
    std::vector<uint8_t> lm_ws8, lm_fixed, cov;
    // where lm_fill_groups writes the per-landmark / per-observation tables: straight into the pinned staging area when it has room (round 5:
    // the tables were filled into the vectors above and copied to the staging area a second time), else into the vectors.  The group-order
    // measurement / weight tables are not built on the host any more: k_lm_tables gathers them on the device (n_meas_*: their sizes)
    bool staged = false;
    int32_t *p_lm_slot = nullptr, *p_lm_ob0 = nullptr, *p_ob_orig = nullptr;
    uint8_t *p_lm_ws8 = nullptr, *p_lm_fixed = nullptr;
    size_t n_lm = 0, n_ob = 0, n_meas_pt = 0, n_meas_ln = 0;
    // scratch of the build
    std::vector<int32_t> kmin, kmax, ord, tmp, cnt, ordall, stamp, span_at, span_end, span_ob0, gcut, bad, c2, pos;
    std::vector<std::pair<int64_t, int32_t>> blk_c, row_c;
    std::vector<uint64_t> kmask;             // keyframe bit mask per landmark
};
}  // namespace plba

struct HostCtx {           // per-problem runtime objects, cached across problems (plba_create / plba_destroy)
    int device = 0;
    hipStream_t stream = nullptr;
    void* h_ctrl = nullptr;      // pinned copy of the control block
    void* h_mail = nullptr;      // mapped, coherent mailbox the decision kernel writes and the host polls
    void* d_mail = nullptr;      // its device address
    plba::StageArea* stage = nullptr;   // pinned upload staging (heap object: HostCtx is copied around by value)
    plba::LmHost* lm_host = nullptr;    // the group tables of the fused passes: megabytes whose pages a fresh problem would fault in again
    plba::StageArea* slide_stage = nullptr;   // pinned staging of plba_slide_window's uploads: its own area, so that the slide need not wait for them before prepare() re-uses `stage`
};


struct plba_problem {
    HostCtx ctx;
    bool have_ctx = false;
    plba_options opt;
    char err[512];
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool ev_sample = false;    // profile = 1: this trial's factorisation span is being timed
    unsigned trial_counter = 0;
    bool spec_lin = false;
    bool lin_in_span = false;
    long prof_lin_launches = 0;     // profile = 2: k_linearize<true> launches whose time went into plba_stats.ms_phase[0], over the problem's life
    std::vector<double> marg_dbg;            // options.diag & PLBA_DIAG_MARG_DUMP: [R, pos, m, n, J (R x pos column-major), r (R)] of the last plba_marginalize* (tools/marg_exact_check.py)
    double marg_path[5] = {0, 0, 0, 0, 0};   // last plba_marginalize*: [0] 0 = block-wise pseudo-inverse, 1 = dense eigen-decomposition of Amm; [1..4] certificate: w_max, smallest kept landmark eigenvalue, tau, smallest pivot
    bool spec_hll = false;
    bool twin_ok = false;      // two-ended multi-launch factorisation of the compact dense system (plba_dense.hip: launch_twin_cholesky)
    plba::TwinView twinv;
    plba::DArr<plba::TwinTile> d_twin_list;
    plba::DArr<int32_t> d_twin_perm, d_twin_xmap, d_twin_fac, d_cs_order;
    plba::DArr<char> d_batch_z, d_batch_u;      // the blocks prepare()'s small buffers are bump-allocated from (plba::DevBatch)
    plba::DevBatch batch;
    // Structure cache (round 5).  Everything prepare() derives from the POSE side of the window — vertex layout, chain elimination maps and
    // buffers, the structural assembly / exchange lists, the band and the multi-chain plan — is a function of a small key (keyframe
    // layout, IMU edge topology, prior vertices, the keyframe co-observation map, options).  Consecutive sliding windows have the same
    // key: the next window then keeps the previous one's device buffers (they live in a batch of their own, `sbatch`, which only starts
    // over on a miss), clears the zero-initialised ones again and skips 0.3 ms of host work per BA call at configs[2].
    struct StructCache {
        bool valid = false;
        std::vector<int32_t> key;
        plba::DevBuf dd;                                   // the dense system's view as the miss left it (per-window fields are refreshed)
        const int32_t *alist = nullptr, *xlist = nullptr, *alist2 = nullptr; int nalist = 0, nxlist = 0, nalist2 = 0;
        const uint8_t* col_gather = nullptr;
        double *Ninv = nullptr, *Nwork = nullptr, *dbgbuf = nullptr;
        bool chain_ok = false;
        std::vector<std::pair<void*, size_t>> zero_log;     // pool buffers the miss zero-filled
        long hits = 0, misses = 0;
    } sc;
    plba::DArr<char> d_sbatch_z, d_sbatch_u;
    plba::DevBatch sbatch;
    plba::DArr<double> d_wtw;          // W^T W tiles of the chain Schur complement (fused landmark path, one GPU: formed in the gather launch)
    plba::DArr<double> d_twin_alt;     // ... and so are its landmark blocks and the pose-side assembly     // the next iteration's linearisation is already in the stream (enqueued behind k_decide)
    // ---- host copy of the uploaded graph -------------------------------------------------------
    bool have_cam = false;
    double fx, fy, cx, cy, Rbc[9], Pbc[3], gw[3] = {0, 0, 0};
    int K = 0, Np = 0, Nl = 0, Ep = 0, El = 0, M = 0;
    std::vector<int32_t> vid_pvr, vid_bias;
    std::vector<double> kf0;              // K x 24 initial records
    std::vector<uint8_t> fix_pvr, fix_bias;
    std::vector<double> lm0;              // L x 6
    std::vector<uint8_t> lm_fixed;        // L
    std::vector<double> pts, lns;
    std::vector<uint8_t> pt_fixed, ln_fixed;
    std::vector<int32_t> po_pt, po_kf, lo_ln, lo_kf;
    std::vector<double> po_uv, po_w, lo_l, lo_w;
    std::vector<uint8_t> level;           // E (points then lines)
    std::vector<int32_t> imu_i, imu_j;
    std::vector<double> imu_pre, imu_ipvr, imu_ibias;
    int pr_n = 0, pr_nv = 0;
    std::vector<int32_t> pr_vid, pr_size, pr_idx;
    std::vector<double> pr_x0, pr_J0, pr_r0;
    plba::Robust rob;
    // shard
    int rank = 0, world = 1;
    plba_allreduce_fn xfn = nullptr;
    void* xuser = nullptr;
    // ---- derived structure ---------------------------------------------------------------------
    bool dirty = true;                    // device image must be rebuilt
    int P = 0, Ppad = 0, ld = 0, L = 0, E = 0;
    std::vector<int32_t> off_pvr, off_bias;
    int cur = 0;                          // index of the current estimate buffers
    // ---- device ----------------------------------------------------------------------------------
    plba::DArr<double> d_kf[2], d_kf_saved, d_lm[2], d_lm_saved;
    // plba_slide_window: the kept landmarks' estimates never leave the device.  d_lm_carry holds the NEW window's landmark array (kept ones
    // gathered from the previous window's current estimates, added ones uploaded); prepare() copies it on the device instead of uploading
    // `pts` / `lns`, whose entries for the kept landmarks are stale while carry_pts / carry_lns are set (plba_set_points / _lines clear them)
    plba::DArr<double> d_lm_carry, d_kf_carry;      // (and the kept keyframes' states: kf0 is stale while carry_kf is set; plba_set_keyframes clears it)
    plba::DArr<int32_t> d_lm_carry_src;
    bool carry_pts = false, carry_lns = false, carry_kf = false;
    // ... and neither do the kept observations' measurements and weights: the slide gathers the new landmark-major arrays on the device
    // (into the *_c buffers: `carry_obs_pending`, swapped in by prepare()); po_uv / po_w / lo_l / lo_w on the host are then stale
    // (carry_po / carry_lo; plba_set_point_obs / plba_set_line_obs clear them)
    plba::DArr<double> d_po_uv_c, d_lo_l_c, d_ob_w_c;
    plba::DArr<double> d_slide_kf, d_slide_lm, d_slide_ob;      // the slide's uploaded additions (members: the gathers that read them are only QUEUED when the slide returns)
    plba::DArr<int32_t> d_obs_carry_src;
    bool carry_po = false, carry_lo = false, carry_obs_pending = false;
    std::vector<int32_t> scr_lm[2], scr_kf[2], scr_src[2];      // the slide's merged lists are built here and swapped with po_pt / po_kf / lo_ln / lo_kf: no fresh pages per slide
    plba::DArr<double> d_po_uv, d_lo_l, d_ob_w, d_ob_chi2, d_erec, d_erec2;
    plba::DArr<int32_t> d_ob_kf, d_ob_slot, d_lm_start, d_off_pvr, d_off_bias;
    plba::DArr<uint8_t> d_level, d_lm_fixed, d_lm_active, d_depth;
    plba::DArr<double> d_hll, d_bl, d_dinv, d_tv, d_xl;
    plba::DArr<int32_t> d_pair_i, d_pair_j, d_pair_start, d_ent_pi, d_ent_pj, d_ent_slot, d_ob_pos;
    plba::DArr<int32_t> d_kfpos, d_ekf, d_ukf, d_trow, d_xlist, d_alist;
    plba::DArr<int32_t> d_esrc, d_esrc_off, d_bkf, d_imu_loc;
    plba::DArr<int> d_trial_cnt;
    plba::DArr<plba::ChunkMeta> d_ch_meta;
    plba::DArr<double> d_schur_part;
    plba::DArr<int> d_pair_cnt;
    std::vector<int32_t> ob_pos;          // keyframe-major record position of every observation
    plba::DArr<int32_t> d_imu_i, d_imu_j;
    plba::DArr<double> d_imu_pre, d_imu_ipvr, d_imu_ibias, d_imu_err, d_imu_chi;
    plba::DArr<int32_t> d_pr_kf, d_pr_isbias, d_pr_size, d_pr_idx, d_pr_x0off, d_pr_off;
    plba::DArr<double> d_pr_x0, d_pr_J0, d_pr_r0, d_pr_err, d_pr_dx, d_pr_chi, d_pr_H;
    plba::DArr<double> d_bprior, d_bprior2;
    plba::DArr<double> d_Hconst, d_Himu, d_bimu, d_Himu2, d_bimu2, d_sys, d_Lfac, d_bpg, d_x, d_Linv, d_LT32, d_rd32;
    plba::DArr<int> d_flow_flags, d_chol_flags;
    int flow_epoch = 0;
    plba::DArr<double> d_chi_part, d_scale_part, d_maxd_part, d_kfdiag, d_red, d_posediag, d_xbuf;
    plba::DArr<plba::Ctrl> d_ctrl;
    plba::DArr<plba_trace_row> d_trace;
    plba::DArr<int> d_trace_n;
    plba::Ctrl* h_ctrl = nullptr;               // pinned
    plba::Mailbox* h_mail = nullptr;            // pinned + device-mapped (k_decide -> host), d_mail = its device address
    plba::Mailbox* d_mail = nullptr;
    unsigned long long mail_seq = 0;
    // chain-variable elimination (options.chain_elim; structure permitting): the dense solver then runs on `dd`
    bool chain_ok = false;
    plba::ChainView cv{};
    plba::DevBuf dd{};
    plba::DArr<int32_t> d_cidx, d_pidx, d_epos, d_seg_start, d_seg_col, d_ppos, d_pslot, d_slotcol;
    plba::DArr<double> d_W, d_Ldinv, d_Lsub, d_sysd, d_Lfacd, d_xd, d_Linvd, d_LT32d, d_rd32d, d_Ninv, d_Ninvd, d_dbgbuf;
    plba::DArr<int> d_flow_flagsd, d_chol_flagsd;
    bool band_ok = false;                       // the compact dense system is banded: twisted in-LDS solver (plba_band.hip)
    plba::BandView bandv{};
    plba::DArr<double> d_band_L, d_band_y, d_band_mid;
    std::vector<int32_t> h_pidx, h_seg_col, h_alist;      // host copies of the chain maps / assembly list (band measurement)
    // fused landmark-major passes (options.lm_fused; plba_lm_dev.h)
    bool lm_ok = false;                         // this upload runs them (structure permitting: chain path, <= 16 observations per landmark — wide groups for 9 .. 16 —, none twice from one keyframe)
    // the landmark state arrays are stored in group order (LmView::lm_grouped): slot -> position (d_lm_pos; the getters, the slide's carry)
    // and position -> slot (d_lm_ord)
    bool lm_grouped = false;
    plba::DArr<int32_t> d_lm_pos, d_lm_ord;
    unsigned long long state_epoch = 1, res_lm_epoch = 0;      // estimates on the device changed | the host mirror of the landmark array is of that epoch
    std::vector<double> res_lm;                 // plba_get_points / plba_get_lines: one read-back per state
    plba::DArr<double> d_lm_pack;               // ... of the PACKED estimates (3 doubles per point, 6 per line: k_lm_pack)
    std::vector<double> lm_hist;                // diagnostics (plba_debug_get "lm_groups")
    int seg_launch_est = 0;                     // dependent factorisation launches the chain segment-length choice expected (diagnostics)
    bool lm_chi_dirty = false;                  // the fused passes' group-order chi2 cache is newer than DevBuf::ob_chi2
    unsigned back_epoch = 0;                    // k_lm_trial launches since the counters were allocated (DevBuf::back_cnt)
    bool lm_disable = false;                    // prepare() found the structure unfit after the fact and rebuilt for the record-based path
    bool lm_spec = false;                       // the accepted trial's Schur pass + gather are already in the stream (enqueued behind the decision)
    plba::LmView lv{};
    plba::DArr<plba::LmGroup> d_lm_grp;
    plba::DArr<int32_t> d_lmg_slot, d_lmg_ob0, d_lmg_orig, d_lmg_blk_ij, d_lmg_blk_start, d_lmg_blk_src, d_lmg_row_kf, d_lmg_row_start, d_lmg_row_src, d_alist2;
    plba::DArr<uint8_t> d_lmg_ws8, d_lmg_fixed, d_lmg_level, d_col_gather;
    plba::DArr<double> d_lmg_meas_pt, d_lmg_meas_ln, d_lmg_wt, d_lmg_chi, d_lmg_part, d_ob_err;
    bool assembled = false;                     // k_landmark_hll already assembled the pose-side system of this iteration
    plba::DevBuf dv;
    std::vector<plba_trace_row> trace;
    bool saved_valid = false;
    hipEvent_t ev[18] = {};               // phase boundaries of one trial (options.profile)
    bool ev_ready = false;
};


// Wait for a stream WITHOUT parking the host thread on an interrupt: hipStreamSynchronize blocks on a completion signal once
// its short spin is over, and on this platform the wake-up was measured at 7-20 ms when other HIP users are alive in the
// process (tools/debug_e2e.py: the 0.5 ms of queued uploads of prepare() then "took" 20 ms).  Polling hipStreamQuery keeps
// the wait in user space; after `spin_ms` it falls back to the blocking call (which also surfaces device errors).
inline hipError_t plba_stream_wait(hipStream_t s, double spin_ms = 50.0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return hipSuccess;
        if (e != hipErrorNotReady) return e;
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > spin_ms) break;
    }
    return hipStreamSynchronize(s);
}

// Copies between device memory and CALLER-OWNED (pageable) host memory go through the context's pinned staging area.
// Handing a pageable pointer to hipMemcpy makes the runtime pin (register) those pages with the GPU for transfers above its
// staging threshold; when the host allocator later gives such pages back to the OS (free of an mmap'ed std::vector / numpy
// buffer, heap trim), the driver's MMU notifier has to invalidate the registration and evicts the process's GPU queues: the
// next stream operations then sat 10-25 ms in the queue (tools/debug_e2e3.py; which call paid depended on the allocator's
// state, e.g. on whether another problem's host vectors were alive).  Pinned memory the library owns never goes back.
inline hipError_t plba_d2h(plba_problem* p, void* dst, const void* src_dev, size_t bytes) {
    if (!bytes) return hipSuccess;
    plba::StageArea* st = p->have_ctx ? p->ctx.stage : nullptr;
    if (!st || !st->base || !st->xfer_cap()) { hipError_t e = hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, p->stream); return e != hipSuccess ? e : plba_stream_wait(p->stream); }
    for (size_t off = 0; off < bytes; off += st->xfer_cap()) {
        const size_t n = std::min(st->xfer_cap(), bytes - off);
        hipError_t e = hipMemcpyAsync(st->xfer(), (const char*)src_dev + off, n, hipMemcpyDeviceToHost, p->stream);
        if (e == hipSuccess) e = plba_stream_wait(p->stream);
        if (e != hipSuccess) return e;
        memcpy((char*)dst + off, st->xfer(), n);
    }
    return hipSuccess;
}
inline hipError_t plba_h2d(plba_problem* p, void* dst_dev, const void* src, size_t bytes) {
    if (!bytes) return hipSuccess;
    plba::StageArea* st = p->have_ctx ? p->ctx.stage : nullptr;
    if (!st || !st->base || !st->xfer_cap()) { hipError_t e = hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, p->stream); return e != hipSuccess ? e : plba_stream_wait(p->stream); }
    for (size_t off = 0; off < bytes; off += st->xfer_cap()) {
        const size_t n = std::min(st->xfer_cap(), bytes - off);
        memcpy(st->xfer(), (const char*)src + off, n);
        hipError_t e = hipMemcpyAsync((char*)dst_dev + off, st->xfer(), n, hipMemcpyHostToDevice, p->stream);
        if (e == hipSuccess) e = plba_stream_wait(p->stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

#define PLBA_FAIL(p, code, ...)                             \
    do {                                                    \
        snprintf((p)->err, sizeof((p)->err), __VA_ARGS__);  \
        return (code);                                      \
    } while (0)
#define PLBA_HIPCK(p, call)                                                                                   \
    do {                                                                                                      \
        hipError_t e__ = (call);                                                                              \
        if (e__ != hipSuccess) PLBA_FAIL(p, PLBA_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
