// libblsw.so — the execution engine and the C ABI of include/blsw.h (host code; the kernels are in k_*.hip).
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <vector>
#include "kcommon.hpp"

using namespace blsw;

namespace {

inline int hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) {
        fprintf(stderr, "[blsw] %s: %s\n", what, hipGetErrorString(e));
        return BLSW_ERR_HIP;
    }
    return BLSW_OK;
}

// RAII: every ABI entry point of an engine runs on the engine's device and restores the caller's
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) hipSetDevice(prev);
    }
};

// device that owns `st` (the current device for the NULL stream): the stateless entry points run there
inline int stream_device(hipStream_t st) {
    int dev = -1;
    if (st) {
        hipDevice_t d;
        if (hipStreamGetDevice(st, &d) == hipSuccess) return (int)d;
    }
    hipGetDevice(&dev);
    return dev;
}

}  // namespace

// Execution engine. Batches ("steps") are SUBMITTED with their input/output pointers and processed in GROUPS of up
// to max_steps batches by one set of launches (N = steps * n lanes per chain kernel), which is what fills the chip:
// one batch of 1024 instances is only 16 waves per chain. Streams (the runtime backs the streams of ONE priority level
// with four hardware queues; streams that share a queue serialise, so the engine uses three levels and few streams):
//   engine-wide, high priority:  sha    : SHA witness bits of every group, in order (needs only msg)  -> ev_sha per group
//                                expand : k_sha_expand per step (bit -> Fp, 31 MB of the 34 MB per instance, the HBM-bound
//                                         kernel), as soon as the group's SHA bits exist: it never waits for the curve /
//                                         pairing chains, so a short job does not pay their latency before its HBM stream
//                                place  : k_place_field per step (staging -> its place around the SHA segment), once the
//                                         group's chains are done and the step's expansion has finished
//                                lat    : the main path of a LATENCY group (a small group that starts a pipeline: launch_group);
//                                         sha and place then also carry the addition chains of its cofactor segments, and the
//                                         buffer's main stream their points (kcommon.hpp: CofactorSide)
//   per group buffer:   normal   main : sha_values -> map -> cofactor -> prepare(H) .......... -> pairing   -> ev_chains
//                       low      aux  : prepare(sig), g1_alloc, g2_alloc  (need only pk / sig)   -> ev_aux
// Field witnesses go to a staging area (coalesced stores). n_buffers group buffers rotate, so the next groups' chains
// overlap the previous groups' placement.
// Materialisation (expand + place of a step into its output) is a queue of jobs in submission order (pump): a free-running
// engine issues a group's jobs when the group is launched; in consumer mode (options.consumer_mode) a job waits until the
// consumer has released its output's previous user, so the 34 MB vectors exist only between expansion and consumption and
// the output ring can be smaller than a group. A step can also leave in compact form (its slices of the staging, copied).
#define BLSW_MAX_BUFFERS 32
#define BLSW_MILLER_CHUNK_DEFAULT 12   // pairs per lane of the pair-parallel Miller product (blsw_verify_multi_batch)
#define BLSW_MILLER_PAR_MIN_PAIRS 8  // below: the serial six-lane team kernel
// clear_cofactor2 on three lanes per pair halves the chain's latency and costs 38 % more products in it: it pays while the launch is
// latency-bound (an 8 192-instance shard in groups of 4: +16 %; one 128-pair instance: 88 -> 60 ms) and costs 1-4 % once the group's
// chains fill the SIMDs (groups of 10 x 1024: profiles/r03_ab_chain_builds.txt section 6)
#define BLSW_COFACTOR_CHUNKED_MAX_LANES 8192
#define BLSW_DEFAULT_EXPAND_VARIANT 0  // 384 x 8: the geometry that stays fast beside every chain build (profiles/r02_ab_fpmul_expand.txt)
#define BLSW_MAX_TIMED 1024
#define BLSW_MAX_CONSUMED 64
struct GroupBuf {
    void* base = nullptr;
    StepDesc* h_desc = nullptr;  // pinned host
    StepDesc* d_desc = nullptr;
    hipStream_t st[2] = {nullptr, nullptr};  // main, aux (a second aux stream for prepare(sig) / the key chains beside the G2 allocation measured slower:
                                             // profiles/r03_ab_chain_builds.txt section 7)
    hipEvent_t ev_start = nullptr, ev_aux = nullptr, ev_sha = nullptr, ev_chains = nullptr, ev_done = nullptr;
    hipEvent_t ev_side = nullptr;
    hipEvent_t ev_cof[BLSW_COFV_EVENTS] = {};  // values-first cofactor chain (kcommon.hpp: CofactorSide): segments, their points, chunks, join
    hipEvent_t* ev_in = nullptr;    // [max_steps] inputs of step s valid (recorded on the submitting stream)
    hipEvent_t* ev_x = nullptr;     // [max_steps] expansion of step s issued and finished
    hipEvent_t* ev_step = nullptr;  // [max_steps] step s complete (witness tensor + results)
    uint64_t first_seq = 0;
    uint32_t steps = 0;
    bool used = false;
    bool lat_group = false;  // the group launched last from this buffer took the latency kernels
    Workspace ws;            // of the group launched last from this buffer
    uint32_t jobs_left = 0;  // its steps whose expansion / placement has not been issued yet
};
struct Job {  // materialisation of one step: bit expansion + field placement into its output (or its compact form)
    int buf;
    uint32_t s;
};
struct blsw_engine {
    uint64_t n = 0;
    uint32_t msg_len = 0, max_steps = 0;
    blsw_layout_t L, LS;
    Modes modes = DEFAULT_MODES;
    blsw_engine_options_t opt;
    int device = -1;
    GroupBuf buf[BLSW_MAX_BUFFERS];
    int nbuf = 0;
    int cur = 0;
    uint32_t pending = 0;
    uint64_t submitted = 0, launched = 0, materialised = 0;
    std::deque<Job> jobs;  // steps whose chains are issued, in submission order, waiting for their output to be free (consumer mode)
    hipStream_t sha = nullptr, expand = nullptr, place = nullptr, lat = nullptr;  // lat: the main path of a latency group
    // HIP event pairs around every k_sha_expand launch since the last stats reset (live roofline measurement)
    hipEvent_t* ev_exp = nullptr;  // 2 * BLSW_MAX_TIMED events
    uint32_t n_timed = 0;
    // consumer releases: output tensor pointer -> event after which it may be overwritten
    const void* consumed_ptr[BLSW_MAX_CONSUMED];
    hipEvent_t consumed_ev[BLSW_MAX_CONSUMED];
    bool consumed_live[BLSW_MAX_CONSUMED];  // a release has been recorded and not yet waited for
    bool held[BLSW_MAX_CONSUMED];           // consumer mode: a step was materialised into this output and it has not been released
    uint32_t refs[BLSW_MAX_CONSUMED];       // consumer mode: accepted steps that will be materialised into this output (slot reserved at submit)
    uint32_t ramp_pos = 0;                  // options.group_ramp: launch groups since creation / the last flush (group sizes 2, 4, 8, ... max_steps)
    bool staged = false;  // false: direct mode (max_steps == 1, no staging; witnesses written in place by the chains)
    bool chains_inlined = false;  // which compilation of the chain kernels (options.chain_variant; kcommon.hpp: BLSW_K)
    uint32_t cofactor_mode = 0;  // clear_cofactor2 with its three chunks on three lanes: 0 by group size, 1 never, 2 always (options.cofactor_mode)
};

static void engine_free(blsw_engine* e) {
    if (!e) return;
    for (int k = 0; k < BLSW_MAX_BUFFERS; k++) {
        GroupBuf& b = e->buf[k];
        if (b.h_desc) hipHostFree(b.h_desc);
        if (b.d_desc) hipFree(b.d_desc);
        for (int i = 0; i < 2; i++)
            if (b.st[i]) hipStreamDestroy(b.st[i]);
        hipEvent_t single[] = {b.ev_start, b.ev_aux, b.ev_sha, b.ev_chains, b.ev_done, b.ev_side};
        for (hipEvent_t ev : single)
            if (ev) hipEventDestroy(ev);
        for (hipEvent_t ev : b.ev_cof)
            if (ev) hipEventDestroy(ev);
        hipEvent_t* arrays[] = {b.ev_in, b.ev_x, b.ev_step};
        for (hipEvent_t* arr : arrays) {
            if (!arr) continue;
            for (uint32_t s = 0; s < e->max_steps; s++)
                if (arr[s]) hipEventDestroy(arr[s]);
            delete[] arr;
        }
    }
    if (e->sha) hipStreamDestroy(e->sha);
    if (e->expand) hipStreamDestroy(e->expand);
    if (e->place) hipStreamDestroy(e->place);
    if (e->lat) hipStreamDestroy(e->lat);
    if (e->ev_exp) {
        for (int i = 0; i < 2 * BLSW_MAX_TIMED; i++)
            if (e->ev_exp[i]) hipEventDestroy(e->ev_exp[i]);
        delete[] e->ev_exp;
    }
    for (int i = 0; i < BLSW_MAX_CONSUMED; i++)
        if (e->consumed_ev[i]) hipEventDestroy(e->consumed_ev[i]);
    delete e;
}

// field witnesses of one step: staged rows (lanes first .. first + n of the tiles at `staging` / the rows at `pair`) -> their
// places around the SHA segment of the step's witness vectors
static void launch_place(blsw_engine* e, hipStream_t st, const Fp* staging, const Fp* pair, uint32_t split_row, uint64_t first, uint64_t* out, uint64_t out_stride) {
    const uint32_t rows = e->L.n_witness - e->L.sha_bits;
    const unsigned chunks = (rows * 3 + 256 * BLSW_PLACE_ITERS - 1) / (256 * BLSW_PLACE_ITERS);
    dim3 grid2(8 * ((chunks + 7) / 8) * (unsigned)e->n);
    hipLaunchKernelGGL(k_place_field, grid2, dim3(256), 0, st, staging, pair, first, e->L.off_expand, e->L.sha_bits, rows, split_row, out, out_stride, (uint32_t)e->n,
                       e->L.off_sig_alloc, e->modes.g2_team ? e->L.off_pk_not_zero - e->L.off_sig_alloc : 0u, e->LS.off_sig_alloc);
}
// N+1-pair product: a step's pair tiles, instance tiles and instance-major rows -> their places in the n instance vectors. Sources: the
// group workspace (first lanes of the step given) or a compact buffer (first lanes 0, instance tiles tile_w wide)
static void launch_place_multi(blsw_engine* e, hipStream_t st, const Workspace& ws, const Fp* pair_tiles, uint64_t first_pair, const Fp* inst_tiles, uint64_t first_inst,
                               uint32_t inst_tile_w, const Fp* rows, uint64_t* out, uint64_t out_stride) {
    const blsw_layout_t& L = e->L;
    const blsw_layout_t& S = e->LS;
    const uint32_t K = L.n_pairs, n = (uint32_t)e->n;
    PlaceRuns pr = {};
    pr.n_runs = 6;
    const uint32_t src[7] = {S.off_msg, S.off_pk_alloc, S.off_pk_not_zero, S.off_map0, S.off_prep_h, S.off_prep_pk, ws.rows_p};
    const uint32_t dst[6] = {L.off_msg, L.off_pk_alloc, L.off_pk_not_zero, L.off_map0, L.off_prep_h, L.off_prep_pk};
    const uint32_t str[6] = {L.stride_msg, L.stride_pk_alloc, L.stride_pk_not_zero, L.stride_hash, L.stride_prep_h, L.stride_prep_pk};
    for (int r = 0; r < 7; r++) pr.src_row[r] = src[r];
    for (int r = 0; r < 6; r++) pr.dst_off[r] = dst[r], pr.dst_stride[r] = str[r];
    auto blocks = [](uint32_t rows_, uint32_t n_y) {
        const unsigned chunks = (rows_ * 3 + 256 * BLSW_PLACE_ITERS - 1) / (256 * BLSW_PLACE_ITERS);
        return dim3(8 * ((chunks + 7) / 8) * n_y);
    };
    hipLaunchKernelGGL(k_place_runs, blocks(ws.rows_p, n * K), dim3(256), 0, st, pair_tiles, first_pair, ws.rows_p, pr, out, out_stride, n * K, K, 64u);
    PlaceRuns pi = {};
    pi.n_runs = 2;
    pi.src_row[0] = S.off_sig_alloc;
    pi.src_row[1] = S.off_prep_sig;
    pi.src_row[2] = ws.rows_i;
    pi.dst_off[0] = L.off_sig_alloc;
    pi.dst_off[1] = L.off_prep_sig;
    hipLaunchKernelGGL(k_place_runs, blocks(ws.rows_i, n), dim3(256), 0, st, inst_tiles, first_inst, ws.rows_i, pi, out, out_stride, n, 1u, inst_tile_w);
    const unsigned chunks = (ws.pair_rows * 3 + 256 * BLSW_PLACE_ITERS - 1) / (256 * BLSW_PLACE_ITERS);
    hipLaunchKernelGGL(k_place_rows, dim3(chunks, n), dim3(256), 0, st, rows, ws.pair_rows, L.off_miller, out, out_stride);
}
static void launch_canonical(blsw_engine* e, hipStream_t st, uint64_t* out, uint64_t out_stride) {
    const uint32_t K = e->L.n_pairs, rows = e->L.n_witness - K * e->L.sha_bits;
    hipLaunchKernelGGL(k_canonical_rows, dim3((rows + 255) / 256, (unsigned)e->n), dim3(256), 0, st, out, out_stride, e->L.off_expand, e->L.sha_bits, rows, K, e->L.stride_hash);
}

static int consumed_slot(blsw_engine* e, const void* ptr) {
    for (int c = 0; c < BLSW_MAX_CONSUMED; c++)
        if (e->consumed_ptr[c] == ptr && (e->consumed_live[c] || e->held[c] || e->refs[c])) return c;
    return -1;
}
// a free slot of the release table; a recorded release whose event has completed needs no wait any more and is recycled
static int free_consumed_slot(blsw_engine* e) {
    for (int c = 0; c < BLSW_MAX_CONSUMED; c++)
        if (!e->consumed_live[c] && !e->held[c] && !e->refs[c]) return c;
    for (int c = 0; c < BLSW_MAX_CONSUMED; c++)
        if (e->consumed_live[c] && !e->held[c] && !e->refs[c] && hipEventQuery(e->consumed_ev[c]) == hipSuccess) {
            e->consumed_live[c] = false;
            return c;
        }
    return -1;
}
// consumer mode: the output a step will be materialised into (its witness tensor, else its compact buffer), or nullptr if untracked
static const void* tracked_output(const blsw_engine* e, const StepDesc& d) {
    const void* ptr = d.out ? static_cast<const void*>(d.out) : d.compact;
    return (e->opt.consumer_mode && e->staged) ? ptr : nullptr;
}
// a consumer's release of an output (blsw_engine_output_consumed): the stream that is about to overwrite it waits for it
static void wait_released(blsw_engine* e, hipStream_t stream, const void* ptr) {
    const int c = consumed_slot(e, ptr);
    if (c >= 0 && e->consumed_live[c]) {
        hipStreamWaitEvent(stream, e->consumed_ev[c], 0);
        e->consumed_live[c] = false;
    }
}
// Issues the expansion (expansion stream: needs the group's SHA bits) and the field placement (placement stream: needs the
// group's chains) of step s of buffer k. A step is complete after both.
static void materialise(blsw_engine* e, int k, uint32_t s) {
    GroupBuf& b = e->buf[k];
    const StepDesc& d = b.h_desc[s];
    const Workspace& ws = b.ws;
    const uint32_t Kc = e->L.n_pairs;
    const CompactForm cf = compact_form(e->n, ws, Kc);
    hipStreamWaitEvent(e->expand, b.ev_sha, 0);
    if (d.compact) {  // the step's bit words leave as they are
        wait_released(e, e->expand, d.compact);
        hipMemcpyAsync(d.compact, ws.bits + (uint64_t)s * (e->n * Kc / 64) * bits_tile_words(ws.sha_words), cf.bits_bytes, hipMemcpyDeviceToDevice, e->expand);
    }
    if (d.out) {
        wait_released(e, e->expand, d.out);
        const bool timed = e->n_timed < BLSW_MAX_TIMED;
        if (timed) hipEventRecord(e->ev_exp[2 * e->n_timed], e->expand);
        const uint32_t K = e->L.n_pairs;  // 1 except for the N+1-pair product: one SHA segment per (instance, pair)
        ExpandArgs xa = {ws.bits, ws.sha_words, (uint64_t)s * e->n * K, e->L.sha_bits, e->L.off_expand, d.out, d.out_stride, K, K > 1 ? e->L.stride_hash : 0u, 0, (int)e->opt.output_form};
        // The expansions of a LATENCY group start when its cofactor segment is done (ev_join), beside prepare(H) and the pairing — two expansions long.
        // Beside the map and cofactor kernels they cost what they hide (BLSW_TRACE_GROUP, 4 x 1024 instances, no profiler attached): the cofactor
        // segment's serial kernels load and store their scratch rows through the write path an expansion saturates (17-20 ms instead of 11-13, also with
        // the expansion dispatched at once as a resident grid, expand_variant 13: 20-28 ms), k_map_q 5.7-9.8 ms instead of 4.4. The pairing kernel
        // gives 5 ms (16.5 instead of 11.4) for 13 ms of expansion. First tensor after 36-39 ms instead of 39-42, shard rate equal within its spread.
        // Consumer mode only: a free-running engine has all of the group's expansions to write, and holding them back is HBM time lost (configs[3],
        // 12 steps: 950 instead of 1 050 instances/s).
        if (b.lat_group && e->opt.consumer_mode) hipStreamWaitEvent(e->expand, b.ev_cof[2 * BLSW_COFV_NSEG + 3], 0);
        const uint32_t variant = e->opt.expand_variant;
        const unsigned lds = e->opt.place_lds;
        launch_expand(variant, e->opt.expand_store, lds, e->expand, xa, (unsigned)(e->n * K));
        if (timed) {
            hipEventRecord(e->ev_exp[2 * e->n_timed + 1], e->expand);
            e->n_timed++;
        }
    }
    hipEventRecord(b.ev_x[s], e->expand);
    hipStreamWaitEvent(e->place, b.ev_chains, 0);
    hipStreamWaitEvent(e->place, b.ev_x[s], 0);
    if (d.compact) {  // and so do its staged field witnesses
        char* dst = reinterpret_cast<char*>(d.compact);
        hipMemcpyAsync(dst + cf.off_staging, ws.staging + (uint64_t)s * (e->n * Kc / 64) * ws.rows_p * 64, cf.staging_bytes, hipMemcpyDeviceToDevice, e->place);
        if (cf.inst_bytes) {  // N+1-pair product: the step's n lanes of the instance tiles (whole tiles, or n lanes of one tile packed [rows_i][n])
            const uint64_t lane0 = (uint64_t)s * e->n;
            const Fp* src = ws.staging_inst + (lane0 >> 6) * (uint64_t)ws.rows_i * 64 + (lane0 & 63);
            if (cf.inst_tile_w == 64)
                hipMemcpyAsync(dst + cf.off_inst, src, cf.inst_bytes, hipMemcpyDeviceToDevice, e->place);
            else
                hipMemcpy2DAsync(dst + cf.off_inst, e->n * sizeof(Fp), src, 64 * sizeof(Fp), e->n * sizeof(Fp), ws.rows_i, hipMemcpyDeviceToDevice, e->place);
        }
        if (cf.pair_bytes) hipMemcpyAsync(dst + cf.off_pair, ws.pair + (uint64_t)s * e->n * ws.pair_rows, cf.pair_bytes, hipMemcpyDeviceToDevice, e->place);
    }
    if (d.out && e->staged) {
        if (e->L.n_pairs > 1)
            launch_place_multi(e, e->place, ws, ws.staging, (uint64_t)s * e->n * e->L.n_pairs, ws.staging_inst, (uint64_t)s * e->n, 64u,
                               ws.pair + (uint64_t)s * e->n * ws.pair_rows, d.out, d.out_stride);
        else
            launch_place(e, e->place, ws.staging, ws.pair, ws.split_row, (uint64_t)s * e->n, d.out, d.out_stride);
    }
    if (d.out && e->opt.output_form) launch_canonical(e, e->place, d.out, d.out_stride);  // direct mode: the chains wrote the rows in place
    hipEventRecord(b.ev_step[s], e->place);
}
// Materialises queued steps in submission order. Free-running engines (consumer_mode 0) issue every step as soon as its
// group is launched; in consumer mode a step whose output still holds an unreleased earlier step stops the queue until
// blsw_engine_output_consumed names that output.
static int pump(blsw_engine* e) {
    while (!e->jobs.empty()) {
        const Job j = e->jobs.front();
        GroupBuf& b = e->buf[j.buf];
        const StepDesc& d = b.h_desc[j.s];
        const void* ptr = tracked_output(e, d);
        int slot = -1;
        if (ptr) {
            slot = consumed_slot(e, ptr);  // reserved when the step was accepted (engine_submit): it exists
            if (slot < 0) return BLSW_ERR_ARG;
            if (e->held[slot]) break;
        }
        materialise(e, j.buf, j.s);
        if (slot >= 0) {
            e->held[slot] = true;
            e->refs[slot]--;
        }
        e->jobs.pop_front();
        e->materialised++;
        if (--b.jobs_left == 0) hipEventRecord(b.ev_done, e->place);
    }
    return hip_ok(hipGetLastError(), "materialise");
}

// BLSW_TRACE_GROUP=1 (diagnostic; not an option of the ABI): timing events after the stages of the K == 1 main stream of every launch group, printed to
// stderr when the engine is destroyed — what a kernel trace shows, without a profiler attached
struct GroupTrace {
    static constexpr int kMarks = 6;
    hipEvent_t ev[kMarks];
    uint64_t lanes;
    bool lat;
};
static std::vector<GroupTrace> g_group_trace;
static bool group_trace_on() {
    static const bool on = getenv("BLSW_TRACE_GROUP") != nullptr;
    return on;
}
static void group_trace_mark(GroupTrace* t, int i, hipStream_t st) {
    if (!t) return;
    hipEventCreate(&t->ev[i]);
    hipEventRecord(t->ev[i], st);
}
static void group_trace_dump() {
    static const char* names[GroupTrace::kMarks - 1] = {"sha_values+map", "cofactor", "prepare(H)", "wait aux", "pairing"};
    for (const GroupTrace& t : g_group_trace) {
        hipEventSynchronize(t.ev[GroupTrace::kMarks - 1]);
        fprintf(stderr, "[blsw group] %llu lanes%s:", (unsigned long long)t.lanes, t.lat ? " (latency kernels)" : "");
        float total = 0;
        for (int i = 0; i + 1 < GroupTrace::kMarks; i++) {
            float ms = 0;
            hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]);
            total += ms;
            fprintf(stderr, " %s %.2f", names[i], ms);
        }
        fprintf(stderr, " | total %.2f ms\n", total);
        for (hipEvent_t ev : t.ev) hipEventDestroy(ev);
    }
    g_group_trace.clear();
}

static int launch_group(blsw_engine* e) {
    GroupBuf& b = e->buf[e->cur];
    const uint32_t steps = e->pending;
    if (steps == 0) return BLSW_OK;
    const uint32_t K = e->L.n_pairs;  // (pk, msg) pairs per instance: 1 except for the N+1-pair product
    Group g;  // per-pair view: one lane per (instance, pair)
    g.N = (uint64_t)steps * e->n * K;
    g.n = (uint32_t)e->n;
    g.K = K;
    g.msg_len = e->msg_len;
    g.desc = b.d_desc;
    g.L = e->L;
    g.LS = e->LS;
    g.ws = carve(b.base, g.N, e->L, e->staged, e->modes, (uint64_t)steps * e->n);
    g.chain_prio = e->opt.prio_mode == 0;
    g.canonical = (int)e->opt.output_form;
    const unsigned g1 = (unsigned)((g.N + 63) / 64);
    const unsigned gt = (unsigned)((g.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE);
    // Which kernels. A SMALL group (at most BLSW_LATENCY_MAX_LANES lanes) that finds the engine's chains idle starts a pipeline: nothing of this engine
    // runs beside it, and its latency — one wave's instruction stream, whatever the group's size — is what a consumer waits for before the first
    // tensors exist. Such a group takes the latency kernels (options.latency_mode 0): map / prepare / G2 allocation on quads and the cofactor chain
    // values first (kcommon.hpp: Latency). "Idle" is the state of the other group buffers' chains, not a count of groups since the last flush.
    // (Letting the second small group of a starting pipeline take them too was measured: 56-57 k against 58-60 k instances/s for an 8 192-instance
    // shard in groups of 4 — two latency groups at once contend for the SIMDs the first one needs.)
    bool idle = true;
    for (int k = 0; k < e->nbuf; k++)
        if (k != e->cur && e->buf[k].used && hipEventQuery(e->buf[k].ev_chains) != hipSuccess) idle = false;
    const bool small = g.ws.cofv != nullptr;  // staged or direct mode (a direct-mode engine is one small group at a time: always latency-bound)
    const uint32_t lm = e->opt.latency_mode;
    Latency lat = {false, false};
    if (small && (lm >= 2 || (lm == 0 && idle))) lat = {lm != 3, lm != 4};
    // with the latency kernels off, such a group takes the inlined compilation of the chain kernels (options.chain_variant 0)
    const bool cold_small = e->opt.chain_variant == 0 && small && idle;
    const ChainKernels ck = chain_kernels(e->chains_inlined || cold_small);
    const bool chunked = e->cofactor_mode == 2 || (e->cofactor_mode == 0 && g.N <= BLSW_COFACTOR_CHUNKED_MAX_LANES);
    // values-first cofactor chain: its per-doubling / per-addition witness phases go to the aux stream, behind the aux chains (enqueued below)
    // ... and the pipelines of its chunks run beside the doubling chain on the buffer's main stream (the segments' points) and the engine's sha and
    // place streams (the addition chains): streams with nothing to do while a cold group's chains run (its SHA bits are enqueued before; placement
    // starts after the chains)
    CofactorSide cof_side;
    cof_side.side = b.st[1];
    cof_side.pts = b.st[0];
    cof_side.acc[0] = e->place;
    cof_side.acc[1] = e->sha;
    for (int i = 0; i < BLSW_COFV_NSEG; i++) cof_side.ev_seg[i] = b.ev_cof[i], cof_side.ev_pts[i] = b.ev_cof[BLSW_COFV_NSEG + i];
    for (int i = 0; i < 3; i++) cof_side.ev_acc[i] = b.ev_cof[2 * BLSW_COFV_NSEG + i];
    cof_side.ev_join = b.ev_cof[2 * BLSW_COFV_NSEG + 3];
    const bool cof_deferred = lat.vf && g.ws.cofv != nullptr;
    // the waves of a latency group's critical path (hash-to-G2, prepare(H), pairing) raise their priority: the streams beside them — the first
    // expansions, the next group's chains, this group's aux chains — have slack, they have none
    Group gm = g;
    if (lat.quad || lat.vf) gm.chain_prio = 1;
    // The main path of a latency group runs on the engine's HIGH-priority latency stream: a pipe of the command processor hands its dispatcher to the
    // highest-priority queue that has a kernel ready, so beside an expansion (high-priority stream, 246 144 workgroups) a kernel launched on a
    // normal-priority stream waits until the expansion has been dispatched to its end (BLSW_TRACE_GROUP: prepare(H) 13 ms instead of 1.8). The
    // buffer's own main stream carries the points of the cofactor segments instead.
    const bool lat_any = lat.quad || lat.vf;
    hipStream_t st = lat_any ? e->lat : b.st[0];
    GroupTrace* trace = nullptr;
    // inputs of every step are ready once its submitting stream reached the point of the submit
    for (uint32_t s = 0; s < steps; s++) hipStreamWaitEvent(st, b.ev_in[s], 0);
    bool any_out = false;
    for (uint32_t s = 0; s < steps; s++) any_out = any_out || b.h_desc[s].out != nullptr || b.h_desc[s].compact != nullptr;
    // direct mode: the chains themselves write into the output tensors, so they wait for the consumer's release
    if (!e->staged)
        for (uint32_t s = 0; s < steps; s++)
            if (b.h_desc[s].out) wait_released(e, st, b.h_desc[s].out);
    hipMemcpyAsync(b.d_desc, b.h_desc, sizeof(StepDesc) * steps, hipMemcpyHostToDevice, st);
    hipEventRecord(b.ev_start, st);
    hipStreamWaitEvent(b.st[1], b.ev_start, 0);
    // sha: the witness bits of the in-circuit SHA-256 (first: the expansion stream is waiting for them)
    hipStreamWaitEvent(e->sha, b.ev_start, 0);
    if (any_out) hipLaunchKernelGGL(ck.sha, dim3(g1), dim3(64), 0, e->sha, g, 1, 0);
    hipEventRecord(b.ev_sha, e->sha);
    if (K > 1) {  // N+1-pair product: per-pair chains on N = steps * n * K lanes, per-signature chains and the Miller product on steps * n
        Group gs = g;
        gs.N = (uint64_t)steps * e->n;
        gs.K = 1;
        hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, gm);
        launch_map(ck, lat, gm, st);
        launch_cofactor(ck, lat, chunked, gm, st, &cof_side);
        launch_prepare(ck, lat, gm, 0, st);
        // aux: what the Miller product waits for (prepare(sig), the keys' prepare_g1) first, then the signature's allocation chain, which only the
        // end of the group waits for (ev_side)
        launch_prepare(ck, lat, gs, 1, b.st[1]);
        hipLaunchKernelGGL(ck.g1, dim3(g1), dim3(64), 0, b.st[1], g);
        hipEventRecord(b.ev_aux, b.st[1]);
        launch_g2_alloc(ck, lat, gs, b.st[1]);
        hipStreamWaitEvent(st, b.ev_aux, 0);
        if (cof_deferred) launch_cofactor_witness(lat, g, cof_side);
        if (K < BLSW_MILLER_PAR_MIN_PAIRS) {
            hipLaunchKernelGGL(k_pairing_team_multi, dim3((unsigned)((gs.N + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, st, gs, K, g.N);
        } else {
            MillerParArgs ma;
            ma.K = K;
            ma.B = BLSW_MILLER_CHUNK_DEFAULT;
            ma.C = (K + ma.B - 1) / ma.B;
            ma.n_h = g.N;
            ma.spine_lane = 0;
            char* p = reinterpret_cast<char*>(b.base) + align_up(g.ws.total_bytes, 256);
            auto take = [&](uint64_t items) {
                Fp* r = reinterpret_cast<Fp*>(p);
                p += align_up(items * 12 * sizeof(Fp), 256);
                return r;
            };
            ma.cprod = take(gs.N * 68 * ma.C);
            ma.q = take(gs.N * 68 * ma.C);
            ma.t = take(gs.N * 68);
            ma.f1 = take(gs.N * 68);
            ma.ffinal = take(gs.N);
            launch_miller_par(gs, ma, st, nullptr, nullptr, nullptr);  // one stream: other groups run beside this one
        }
    } else {
        // main, first part: the hash-to-G2 critical path
        if (group_trace_on()) {
            g_group_trace.push_back(GroupTrace{{}, g.N, lat.quad || lat.vf});
            trace = &g_group_trace.back();
        }
        group_trace_mark(trace, 0, st);
        hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, gm);
        launch_map(ck, lat, gm, st);
        group_trace_mark(trace, 1, st);
        launch_cofactor(ck, lat, chunked, gm, st, &cof_side);
        group_trace_mark(trace, 2, st);
        launch_prepare(ck, lat, gm, 0, st);
        group_trace_mark(trace, 3, st);
        // aux: prepare_g2(sig) and the key's allocation + prepare_g1 — what the pairing waits for (ev_aux) — then the signature's allocation chain, which
        // only the end of the group waits for (ev_side): the longest aux kernel no longer delays the pairing of a latency-bound group
        hipStream_t sb = b.st[1];
        launch_prepare(ck, lat, g, 1, sb);
        if (e->L.n_keys) {  // aggregate_verify: one lane per (instance, key) allocates, then mapped_aggregate + pk != 0 + prepare_g1 per instance
            hipLaunchKernelGGL(ck.agg_keys, dim3((unsigned)((g.N * e->L.n_keys + 63) / 64)), dim3(64), 0, sb, g, g.ws.keyproj);
            hipLaunchKernelGGL(ck.agg_sum, dim3(g1), dim3(64), 0, sb, g, (const Fp*)g.ws.keyproj);
        } else  // params_mode: lanes [N, 2 N) allocate and prepare the generator (k_g1)
            hipLaunchKernelGGL(ck.g1, dim3(e->L.params_mode ? (unsigned)((2 * g.N + 63) / 64) : g1), dim3(64), 0, sb, g);
        hipEventRecord(b.ev_aux, sb);
        if (e->L.sig_mode) {
            // SignatureVar::new_variable(Input): no allocation chain (prepare(sig) wrote the instance variables)
        } else if (e->modes.g2_team)
            hipLaunchKernelGGL(k_g2_alloc_team, dim3(gt), dim3(64), 0, b.st[1], g);
        else
            launch_g2_alloc(ck, lat, g, b.st[1]);
    }
    if (K == 1) {
        if (cof_deferred) launch_cofactor_witness(lat, g, cof_side);
        // main, second part: the pairing
        hipStreamWaitEvent(st, b.ev_aux, 0);
        group_trace_mark(trace, 4, st);
        launch_pairing(gm, e->modes, st);
        group_trace_mark(trace, 5, st);
    }
    // the group's chains are done when the aux stream's tail (G2 allocation, deferred witness phases of the cofactor chain) is
    hipEventRecord(b.ev_side, b.st[1]);
    hipStreamWaitEvent(st, b.ev_side, 0);
    hipStreamWaitEvent(st, b.ev_sha, 0);
    hipEventRecord(b.ev_chains, st);
    // expansion + placement of the group's steps: queued, issued in submission order (at once unless a consumer holds an output)
    b.ws = g.ws;
    b.jobs_left = steps;
    for (uint32_t s = 0; s < steps; s++) e->jobs.push_back({e->cur, s});
    b.used = true;
    b.lat_group = lat.quad || lat.vf;
    b.first_seq = e->launched;
    b.steps = steps;
    e->launched += steps;
    e->pending = 0;
    if (e->ramp_pos < 30) e->ramp_pos++;
    e->cur = (e->cur + 1) % e->nbuf;
    if (hip_ok(hipGetLastError(), "launch")) return BLSW_ERR_HIP;
    return pump(e);
}

// Side streams of a DIRECT call's values-first cofactor chain (kcommon.hpp: CofactorSide), created once per host thread and device and kept (an event
// is re-recorded per call; a wait refers to the record that preceded it). Without them the segments' phases run one after the other on the caller's stream.
struct DirectLanes {
    int device = -1;
    hipStream_t st[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[BLSW_COFV_EVENTS + 1] = {};
    bool ok = false;
};
static DirectLanes* direct_lanes(int dev) {
    static thread_local std::map<int, DirectLanes> lanes;
    DirectLanes& d = lanes[dev];
    if (d.device != dev) {
        d.device = dev;
        d.ok = true;
        for (hipStream_t& q : d.st) d.ok = d.ok && hipStreamCreateWithFlags(&q, hipStreamNonBlocking) == hipSuccess;
        for (hipEvent_t& ev : d.ev) d.ok = d.ok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
    }
    return d.ok ? &d : nullptr;
}
// launch_cofactor of a direct call on `st`, with its pipelines on the thread's side streams; `st` has waited for all of it on return
static void launch_cofactor_direct(const ChainKernels& ck, Latency lat, const Group& g, hipStream_t st, int dev) {
    const bool chunked = g.N <= BLSW_COFACTOR_CHUNKED_MAX_LANES;
    DirectLanes* d = (lat.vf && g.ws.cofv) ? direct_lanes(dev) : nullptr;
    if (!d) {
        launch_cofactor(ck, lat, chunked, g, st);
        return;
    }
    CofactorSide cs;
    cs.side = d->st[0];
    cs.pts = d->st[1];
    cs.acc[0] = d->st[2];
    cs.acc[1] = d->st[3];
    for (int i = 0; i < BLSW_COFV_NSEG; i++) cs.ev_seg[i] = d->ev[i], cs.ev_pts[i] = d->ev[BLSW_COFV_NSEG + i];
    for (int i = 0; i < 3; i++) cs.ev_acc[i] = d->ev[2 * BLSW_COFV_NSEG + i];
    cs.ev_join = d->ev[2 * BLSW_COFV_NSEG + 3];
    launch_cofactor(ck, lat, chunked, g, st, &cs);
    launch_cofactor_witness(lat, g, cs);
    hipEventRecord(d->ev[BLSW_COFV_EVENTS], cs.side);
    hipStreamWaitEvent(st, d->ev[BLSW_COFV_EVENTS], 0);
}

extern "C" {

int blsw_version(void) { return BLSW_ABI_VERSION; }

int blsw_layout(uint32_t msg_len, blsw_layout_t* out) {
    if (!out || msg_len > 65535) return BLSW_ERR_ARG;
    make_layout(msg_len, out);
    return BLSW_OK;
}

int blsw_layout_params(uint32_t msg_len, uint32_t params_mode, blsw_layout_t* out) {
    if (!out || msg_len > 65535 || params_mode > 1) return BLSW_ERR_ARG;
    make_layout(msg_len, out, 0, 1, params_mode == 1);
    return BLSW_OK;
}

int blsw_layout_io(uint32_t msg_len, uint32_t pk_mode, uint32_t sig_mode, blsw_layout_t* out) {
    if (!out || msg_len > 65535 || pk_mode > 1 || sig_mode > 1) return BLSW_ERR_ARG;
    make_layout(msg_len, out, 0, 1, false, pk_mode == 1, sig_mode == 1);
    return BLSW_OK;
}

int blsw_engine_options_default(blsw_engine_options_t* o) {
    if (!o) return BLSW_ERR_ARG;
    o->device = -1;
    o->n_keys = 0;
    o->pairing_mode = 0;
    o->g2_mode = 0;
    o->expand_variant = BLSW_DEFAULT_EXPAND_VARIANT;
    o->expand_store = 0;  // plain stores: nontemporal ones cost 8-10 % since the chains' stack traffic was cut
    o->prio_mode = 1;
    o->place_lds = 0;
    o->consumer_mode = 0;
    o->output_form = 0;
    o->chain_variant = 0;
    o->n_pairs = 0;
    o->cofactor_mode = 0;
    o->params_mode = 0;
    o->group_ramp = 0;
    o->latency_mode = 0;
    o->pk_mode = 0;
    o->sig_mode = 0;
    return BLSW_OK;
}

int blsw_engine_workspace_bytes_ex(uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, const blsw_engine_options_t* options, uint64_t* bytes) {
    if (!bytes || n == 0 || max_steps == 0 || n_buffers == 0 || n_buffers > BLSW_MAX_BUFFERS || msg_len > 65535 || !options || options->n_keys > 65535 ||
        options->n_pairs > 4096 || (options->n_pairs > 1 && options->n_keys) || options->params_mode > 1 ||
        (options->params_mode && (options->n_keys || options->n_pairs > 1 || options->pairing_mode)) || options->pk_mode > 1 || options->sig_mode > 1 ||
        ((options->pk_mode || options->sig_mode) && (options->n_keys || options->n_pairs > 1 || options->params_mode || options->g2_mode)))
        return BLSW_ERR_ARG;
    blsw_layout_t L;
    const uint32_t K = options->n_pairs > 1 ? options->n_pairs : 1;
    make_layout(msg_len, &L, options->n_keys, K, options->params_mode == 1, options->pk_mode == 1, options->sig_mode == 1);
    const bool staged = max_steps > 1 || n_buffers > 1;
    // the same workspace serves every kernel variant: the largest carve of the three mode combinations
    uint64_t need = 0;
    const Modes all[3] = {{true, false}, {true, true}, {false, false}};
    // a launch group may be any number of pending batches up to max_steps, and groups of at most BLSW_LATENCY_MAX_LANES lanes carry the
    // scratch of the latency kernels: the largest such group can need more than the largest group
    const uint64_t small_steps = BLSW_LATENCY_MAX_LANES / (n * K) < max_steps ? BLSW_LATENCY_MAX_LANES / (n * K) : max_steps;
    for (const Modes& m : all) {
        uint64_t t = carve(nullptr, n * max_steps * K, L, staged, m, n * max_steps).total_bytes;
        need = t > need ? t : need;
        if (small_steps) {
            t = carve(nullptr, n * small_steps * K, L, staged, m, n * small_steps).total_bytes;
            need = t > need ? t : need;
        }
    }
    if (K > 1) need = align_up(need, 256) + miller_par_bytes(n * max_steps, K, BLSW_MILLER_CHUNK_DEFAULT);  // value stores of the pair-parallel Miller product
    *bytes = (uint64_t)n_buffers * align_up(need, 4096);
    return BLSW_OK;
}
int blsw_engine_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, uint64_t* bytes) {
    blsw_engine_options_t o;
    blsw_engine_options_default(&o);
    return blsw_engine_workspace_bytes_ex(n, msg_len, max_steps, n_buffers, &o, bytes);
}

int blsw_engine_create_ex(blsw_engine_t** out, uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, const blsw_engine_options_t* options,
                          void* d_workspace, uint64_t workspace_bytes) {
    // n is the y extent of the expansion / canonical-form launches (one row of workgroups per instance): at most 65535
    if (!out || n == 0 || n > 65535 || max_steps == 0 || !d_workspace || n_buffers == 0 || n_buffers > BLSW_MAX_BUFFERS || !options || msg_len > 65535)
        return BLSW_ERR_ARG;
    // consumer mode is late materialisation out of the staging: a direct-mode engine (one step, one buffer) writes its witnesses in
    // place while the chains run and could not honour a held output
    if (options->consumer_mode > 1 || (options->consumer_mode == 1 && max_steps == 1 && n_buffers == 1)) return BLSW_ERR_ARG;
    if (options->pairing_mode > 1 || options->g2_mode > 1 || (options->g2_mode == 1 && options->pairing_mode != 0) || options->expand_store > 3 ||
        options->prio_mode > 2 || options->group_ramp > 1 || options->latency_mode > 4 || options->output_form > 1 || options->chain_variant > 2 || options->cofactor_mode > 2 || (options->expand_variant & 0xff) > 13 || (options->expand_variant >> 9) || options->n_keys > 65535 ||
        (options->n_keys && options->g2_mode) || options->n_pairs > 4096)
        return BLSW_ERR_ARG;
    // N+1-pair product (options.n_pairs = K > 1): a staged engine with the default kernel modes; its expansion launch has one row of
    // workgroups per (instance, pair)
    if (options->n_pairs > 1 && (options->n_keys || options->pairing_mode || options->g2_mode || !(max_steps > 1 || n_buffers > 1) || n * options->n_pairs > 65535))
        return BLSW_ERR_ARG;
    // ParametersVar allocated as witnesses: the single-key circuit with the six-lane pairing kernel (k_pairing_team_pv)
    if (options->params_mode > 1 || (options->params_mode && (options->n_keys || options->n_pairs > 1 || options->pairing_mode))) return BLSW_ERR_ARG;
    // PublicKeyVar / SignatureVar allocated as public inputs: the single-key circuit with Constant parameters and the one-lane G2 kernels
    if (options->pk_mode > 1 || options->sig_mode > 1 ||
        ((options->pk_mode || options->sig_mode) && (options->n_keys || options->n_pairs > 1 || options->params_mode || options->g2_mode)))
        return BLSW_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return BLSW_ERR_NO_DEVICE;
    int dev = options->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return BLSW_ERR_NO_DEVICE;
    if (dev >= ndev) return BLSW_ERR_ARG;
    DeviceGuard guard(dev);
    uint64_t need = 0;
    if (blsw_engine_workspace_bytes_ex(n, msg_len, max_steps, n_buffers, options, &need)) return BLSW_ERR_ARG;
    if (workspace_bytes < need) return BLSW_ERR_WORKSPACE;
    // Scratch guard. ROCr backs a queue's scratch for full-device occupancy: stack bytes per lane x 64 lanes x wave slots
    // (CUs x 32), per queue that runs the kernel. The single-lane pairing kernel (9.7 KB of stack) on four or more group
    // buffers made the runtime abort with HSA_STATUS_ERROR_OUT_OF_RESOURCES; refuse instead.
    if (options->pairing_mode == 1) {
        hipFuncAttributes fa;
        int cus = 256;
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        uint64_t stack = 10240;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_pairing)) == hipSuccess && fa.localSizeBytes) stack = fa.localSizeBytes;
        const uint64_t projected = stack * 64ull * (uint64_t)cus * 32ull * n_buffers;
        if (projected > (16ull << 30)) {
            fprintf(stderr, "[blsw] pairing_mode 1 with %u group buffers needs about %.1f GB of per-queue scratch: refused (use pairing_mode 0 or n_buffers <= 3)\n",
                    n_buffers, projected / 1e9);
            return BLSW_ERR_SCRATCH;
        }
    }
    blsw_engine* e = new blsw_engine();
    e->n = n;
    e->msg_len = msg_len;
    e->max_steps = max_steps;
    e->opt = *options;
    e->opt.device = dev;
    e->device = dev;
    e->modes = {options->pairing_mode == 0, options->g2_mode == 1};
    e->staged = max_steps > 1 || n_buffers > 1;
    e->cofactor_mode = options->cofactor_mode;
    e->chains_inlined = options->chain_variant == 2 || (options->chain_variant == 0 && !e->staged);
    make_layout(msg_len, &e->L, options->n_keys, options->n_pairs > 1 ? options->n_pairs : 1, options->params_mode == 1, options->pk_mode == 1, options->sig_mode == 1);
    e->LS = e->L.n_pairs > 1 ? staging_layout_multi(e->L).LS : staging_layout(e->L, e->modes);
    for (int i = 0; i < BLSW_MAX_CONSUMED; i++) {
        e->consumed_ptr[i] = nullptr;
        e->consumed_ev[i] = nullptr;
        e->consumed_live[i] = false;
        e->held[i] = false;
        e->refs[i] = 0;
    }
    e->nbuf = (int)n_buffers;
    int rc = BLSW_OK;
    auto chk = [&](hipError_t err, const char* what) {
        if (rc == BLSW_OK && hip_ok(err, what)) rc = BLSW_ERR_HIP;
        return rc == BLSW_OK;
    };
    e->ev_exp = new hipEvent_t[2 * BLSW_MAX_TIMED]();
    for (int i = 0; i < 2 * BLSW_MAX_TIMED && rc == BLSW_OK; i++) chk(hipEventCreate(&e->ev_exp[i]), "event create");
    int prio_lo = 0, prio_hi = 0;
    chk(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi), "priority range");  // (least, greatest): numerically lower = higher priority
    // three priority levels = three pools of hardware queues: [placement, main chains, aux chains] from high to low
    // (prio_mode 1, default), [main, aux, placement] (prio_mode 0) or everything at the middle level (prio_mode 2)
    const int prio_mid = (prio_hi + 1 <= prio_lo) ? prio_hi + 1 : prio_lo;  // -1 high, 0 normal, 1 low on this runtime
    int place_prio = prio_hi, main_prio = prio_mid, aux_prio = prio_lo;
    if (e->opt.prio_mode == 0) place_prio = prio_lo, main_prio = prio_hi, aux_prio = prio_mid;
    if (e->opt.prio_mode == 2) place_prio = main_prio = aux_prio = prio_mid;
    for (int k = 0; k < e->nbuf && rc == BLSW_OK; k++) {
        GroupBuf& b = e->buf[k];
        b.base = reinterpret_cast<char*>(d_workspace) + (uint64_t)k * (need / e->nbuf);
        chk(hipHostMalloc(reinterpret_cast<void**>(&b.h_desc), sizeof(StepDesc) * max_steps, hipHostMallocDefault), "host alloc");
        chk(hipMalloc(reinterpret_cast<void**>(&b.d_desc), sizeof(StepDesc) * max_steps), "desc alloc");
        chk(hipStreamCreateWithPriority(&b.st[0], hipStreamNonBlocking, main_prio), "stream create");
        chk(hipStreamCreateWithPriority(&b.st[1], hipStreamNonBlocking, aux_prio), "stream create");
        hipEvent_t* single[] = {&b.ev_start, &b.ev_aux, &b.ev_sha, &b.ev_chains, &b.ev_done, &b.ev_side};
        for (hipEvent_t* ev : single) chk(hipEventCreateWithFlags(ev, hipEventDisableTiming), "event create");
        for (hipEvent_t& ev : b.ev_cof) chk(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "event create");
        b.ev_in = new hipEvent_t[max_steps]();
        b.ev_x = new hipEvent_t[max_steps]();
        b.ev_step = new hipEvent_t[max_steps]();
        for (uint32_t s = 0; s < max_steps && rc == BLSW_OK; s++) {
            chk(hipEventCreateWithFlags(&b.ev_in[s], hipEventDisableTiming), "event create");
            chk(hipEventCreateWithFlags(&b.ev_x[s], hipEventDisableTiming), "event create");
            chk(hipEventCreateWithFlags(&b.ev_step[s], hipEventDisableTiming), "event create");
        }
    }
    chk(hipStreamCreateWithPriority(&e->sha, hipStreamNonBlocking, place_prio), "stream create");
    chk(hipStreamCreateWithPriority(&e->expand, hipStreamNonBlocking, place_prio), "stream create");
    chk(hipStreamCreateWithPriority(&e->place, hipStreamNonBlocking, place_prio), "stream create");
    chk(hipStreamCreateWithPriority(&e->lat, hipStreamNonBlocking, place_prio), "stream create");
    // Scratch pre-warm. The pairing kernel has the largest stack (4.5 KB per lane): the first launch of a full-size group on
    // a queue makes the runtime grow that queue's scratch, which stalls the queue for ~60 ms (measured: the pairing of the
    // first group of every buffer started 60 ms late). One launch of the same grid with N = 0 (every wave exits at once)
    // per main stream pays that here instead of in the caller's first groups.
    if (rc == BLSW_OK && e->L.n_pairs == 1) {
        Group g0;
        memset(&g0, 0, sizeof(g0));
        g0.n = (uint32_t)n;
        g0.K = 1;
        const uint64_t Nmax = n * max_steps;
        for (int k = 0; k < e->nbuf; k++) {
            if (e->modes.pairing_team)
                hipLaunchKernelGGL(k_pairing_team, dim3((unsigned)((Nmax + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, e->buf[k].st[0], g0);
            else
                hipLaunchKernelGGL(k_pairing, dim3((unsigned)((Nmax + 63) / 64)), dim3(64), 0, e->buf[k].st[0], g0);
        }
        for (int k = 0; k < e->nbuf; k++) chk(hipStreamSynchronize(e->buf[k].st[0]), "scratch pre-warm");
    }
    if (rc != BLSW_OK) {
        engine_free(e);
        return rc;
    }
    *out = e;
    return BLSW_OK;
}

int blsw_engine_create(blsw_engine_t** out, uint64_t n, uint32_t msg_len, uint32_t max_steps, uint32_t n_buffers, void* d_workspace,
                       uint64_t workspace_bytes) {
    blsw_engine_options_t o;
    blsw_engine_options_default(&o);
    return blsw_engine_create_ex(out, n, msg_len, max_steps, n_buffers, &o, d_workspace, workspace_bytes);
}

int blsw_engine_destroy(blsw_engine_t* e) {
    if (!e) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    hipDeviceSynchronize();
    if (group_trace_on()) group_trace_dump();
    engine_free(e);
    return BLSW_OK;
}

// Can a step with this output be accepted now? BLSW_ERR_BUSY (nothing queued, nothing issued) when the buffer's previous group still
// has steps waiting for their outputs, or when the release table has no slot for a new output (BLSW_MAX_CONSUMED distinct outputs
// held, reserved or with an unfinished release: the caller drains / releases and calls again).
static int submit_admissible(blsw_engine* e, const void* tracked) {
    GroupBuf& b = e->buf[e->cur];
    if (e->pending == 0 && b.used && b.jobs_left) return BLSW_ERR_BUSY;
    if (tracked && consumed_slot(e, tracked) < 0 && free_consumed_slot(e) < 0) return BLSW_ERR_BUSY;
    return BLSW_OK;
}
// steps of the next launch group: max_steps, or with options.group_ramp 2, 4, 8, ... up to max_steps for the first groups after creation /
// a flush (kept as an option; measured useless: a group's chain latency is one wave's latency, 59 ms for 2 x 1024 instances and 63 ms for 4 x 1024)
static uint32_t group_target(const blsw_engine* e) {
    if (!e->opt.group_ramp || e->ramp_pos >= 30) return e->max_steps;
    const uint32_t t = 2u << e->ramp_pos;
    return t < e->max_steps ? t : e->max_steps;
}
static int engine_submit(blsw_engine_t* e, const StepDesc& step, void* stream_) {
    if (step.out && step.out_stride < e->L.n_witness) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    GroupBuf& b = e->buf[e->cur];
    const void* tracked = tracked_output(e, step);
    // the buffer's previous group must have been fully placed before its staging is overwritten; in consumer mode some of its steps
    // may still wait for their outputs: the caller has to drain (wait_step / output_consumed) first. Checked before anything is taken.
    if (int rc = submit_admissible(e, tracked)) return rc;
    if (e->pending == 0 && b.used) {
        if (hip_ok(hipEventSynchronize(b.ev_done), "event sync")) return BLSW_ERR_HIP;
        b.used = false;
    }
    if (hip_ok(hipEventRecord(b.ev_in[e->pending], reinterpret_cast<hipStream_t>(stream_)), "event record")) return BLSW_ERR_HIP;
    if (tracked) {  // reserve the output's slot of the release table: pump() can no longer run out of slots for an accepted step
        int slot = consumed_slot(e, tracked);
        if (slot < 0) {
            slot = free_consumed_slot(e);
            e->consumed_ptr[slot] = tracked;
            e->consumed_live[slot] = false;
        }
        e->refs[slot]++;
    }
    b.h_desc[e->pending] = step;
    e->pending++;
    e->submitted++;
    if (e->pending >= group_target(e)) return launch_group(e);
    return BLSW_OK;
}
int blsw_engine_submit(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, uint64_t* d_witness,
                       uint64_t witness_stride, int32_t* d_result, void* stream_) {
    if (!e || e->L.n_keys || e->L.n_pairs > 1 || !d_pk_xy || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {d_pk_xy, d_sig_xy, d_msg, d_witness, witness_stride, d_result, nullptr, nullptr, nullptr, nullptr, nullptr};
    return engine_submit(e, d, stream_);
}
int blsw_engine_submit_io(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, uint64_t* d_instance, uint64_t* d_witness,
                          uint64_t witness_stride, int32_t* d_result, void* stream_) {
    if (!e || e->L.n_keys || e->L.n_pairs > 1 || e->L.params_mode || !d_pk_xy || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {d_pk_xy, d_sig_xy, d_msg, d_witness, witness_stride, d_result, nullptr, nullptr, nullptr, nullptr, nullptr, d_instance};
    return engine_submit(e, d, stream_);
}
// N+1-pair product through the engine (an engine created with options.n_pairs = K): one batch of n instances, each ONE signature
// over K (pk, msg) pairs; same grouping / staging / streaming placement / consumer mode as blsw_engine_submit
int blsw_engine_submit_multi(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_msgs, const uint64_t* d_sig_xy, uint64_t* d_witness, uint64_t witness_stride,
                             int32_t* d_result, void* stream_) {
    if (!e || e->L.n_pairs < 2 || !d_pks_xy || !d_sig_xy || (!d_msgs && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {d_pks_xy, d_sig_xy, d_msgs, d_witness, witness_stride, d_result, nullptr, nullptr, nullptr, nullptr, nullptr};
    return engine_submit(e, d, stream_);
}
// One call from compressed bytes: decode on `stream`, then the step; result[i] = gadget Boolean AND both points decoded
// to non-identity subgroup points (tests/tests.rs:244-263: a point that fails to decode is replaced by the default — the identity —
// and the case must come out false). The decoded coordinates live in caller buffers (they are the step's inputs).
int blsw_engine_submit_bytes(blsw_engine_t* e, const uint8_t* d_pk48, const uint8_t* d_sig96, const uint8_t* d_msg, uint64_t* d_pk_xy, uint64_t* d_sig_xy,
                             int32_t* d_status, uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, void* stream_) {
    if (!e || e->L.n_keys || e->L.n_pairs > 1 || !d_pk48 || !d_sig96 || !d_pk_xy || !d_sig_xy || !d_status || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    if (d_witness && witness_stride < e->L.n_witness) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    // the decode is issued only if the step can be taken (same conditions as engine_submit: nothing may be half done on BUSY)
    if (int rc = submit_admissible(e, (e->opt.consumer_mode && e->staged) ? static_cast<const void*>(d_witness) : nullptr)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_decode, dim3((unsigned)((2 * e->n + 63) / 64)), dim3(64), 0, st, d_pk48, d_sig96, e->n, d_pk_xy, d_sig_xy, d_status);
    if (hip_ok(hipGetLastError(), "launch")) return BLSW_ERR_HIP;
    StepDesc d = {d_pk_xy, d_sig_xy, d_msg, d_witness, witness_stride, d_result, nullptr, nullptr, nullptr, nullptr, d_status};
    return engine_submit(e, d, stream_);
}
// aggregate_verify through the engine (an engine created with options.n_keys = K): one batch of n instances of K keys each
int blsw_engine_submit_aggregate(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_bitmap, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                 uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, uint32_t* d_count, void* stream_) {
    if (!e || !e->L.n_keys || !d_pks_xy || !d_bitmap || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {nullptr, d_sig_xy, d_msg, d_witness, witness_stride, d_result, d_pks_xy, d_bitmap, d_count, nullptr};
    return engine_submit(e, d, stream_);
}

// Compact wire form (the all-gather of full witness vectors is capped by xGMI at a fraction of the generation
// rate; 2.6 MB per instance travel instead of 34 MB and the receiver expands them).
int blsw_engine_compact_bytes(blsw_engine_t* e, uint64_t* bytes) {
    if (!e || !bytes || !e->staged || !compact_shape_ok(e->n, e->L.n_pairs)) return BLSW_ERR_ARG;
    *bytes = compact_form(e->n, carve(nullptr, e->n * e->L.n_pairs, e->L, true, e->modes, e->n), e->L.n_pairs).total;
    return BLSW_OK;
}
int blsw_engine_submit_compact(blsw_engine_t* e, const uint64_t* d_pk_xy, const uint64_t* d_sig_xy, const uint8_t* d_msg, void* d_compact, int32_t* d_result,
                               void* stream_) {
    if (!e || e->L.n_keys || e->L.n_pairs > 1 || !e->staged || e->n % 64 || !d_compact || !d_pk_xy || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {d_pk_xy, d_sig_xy, d_msg, nullptr, 0, d_result, nullptr, nullptr, nullptr, d_compact};
    return engine_submit(e, d, stream_);
}
int blsw_engine_submit_aggregate_compact(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_bitmap, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                         void* d_compact, int32_t* d_result, uint32_t* d_count, void* stream_) {
    if (!e || !e->L.n_keys || !e->staged || e->n % 64 || !d_compact || !d_pks_xy || !d_bitmap || !d_sig_xy || (!d_msg && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {nullptr, d_sig_xy, d_msg, nullptr, 0, d_result, d_pks_xy, d_bitmap, d_count, d_compact};
    return engine_submit(e, d, stream_);
}
// the N+1-pair product's step in compact form (engine created with options.n_pairs = K; n K a multiple of 64, n a divisor or multiple of 64)
int blsw_engine_submit_multi_compact(blsw_engine_t* e, const uint64_t* d_pks_xy, const uint8_t* d_msgs, const uint64_t* d_sig_xy, void* d_compact, int32_t* d_result,
                                     void* stream_) {
    if (!e || e->L.n_pairs < 2 || !compact_shape_ok(e->n, e->L.n_pairs) || !d_compact || !d_pks_xy || !d_sig_xy || (!d_msgs && e->msg_len)) return BLSW_ERR_ARG;
    StepDesc d = {d_pks_xy, d_sig_xy, d_msgs, nullptr, 0, d_result, nullptr, nullptr, nullptr, d_compact, nullptr};
    return engine_submit(e, d, stream_);
}
// receiver side: one batch in compact form -> its n witness vectors, on `stream` (the expansion and placement kernels of the
// engine's own steps, pointed at the compact buffer)
int blsw_engine_expand_compact(blsw_engine_t* e, const void* d_compact, uint64_t* d_witness, uint64_t witness_stride, void* stream_) {
    if (!e || !e->staged || !compact_shape_ok(e->n, e->L.n_pairs) || !d_compact || !d_witness || witness_stride < e->L.n_witness) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    const uint32_t K = e->L.n_pairs;
    const Workspace w = carve(nullptr, e->n * K, e->L, true, e->modes, e->n);
    const CompactForm cf = compact_form(e->n, w, K);
    const char* src = reinterpret_cast<const char*>(d_compact);
    if (K > 1) {
        ExpandArgs xm = {reinterpret_cast<const uint32_t*>(src), w.sha_words, 0, e->L.sha_bits, e->L.off_expand, d_witness, witness_stride, K, e->L.stride_hash, 0, (int)e->opt.output_form};
        launch_expand(e->opt.expand_variant, e->opt.expand_store, e->opt.place_lds, st, xm, (unsigned)(e->n * K));
        launch_place_multi(e, st, w, reinterpret_cast<const Fp*>(src + cf.off_staging), 0, reinterpret_cast<const Fp*>(src + cf.off_inst), 0, cf.inst_tile_w,
                           reinterpret_cast<const Fp*>(src + cf.off_pair), d_witness, witness_stride);
        if (e->opt.output_form) launch_canonical(e, st, d_witness, witness_stride);
        return hip_ok(hipGetLastError(), "expand compact");
    }
    ExpandArgs xa = {reinterpret_cast<const uint32_t*>(src), w.sha_words, 0, e->L.sha_bits, e->L.off_expand, d_witness, witness_stride, 1u, 0u, 0, (int)e->opt.output_form};
    launch_expand(e->opt.expand_variant, e->opt.expand_store, e->opt.place_lds, st, xa, (unsigned)e->n);
    launch_place(e, st, reinterpret_cast<const Fp*>(src + cf.off_staging), reinterpret_cast<const Fp*>(src + cf.off_pair), w.split_row, 0, d_witness, witness_stride);
    if (e->opt.output_form) launch_canonical(e, st, d_witness, witness_stride);
    return hip_ok(hipGetLastError(), "expand compact");
}

// launches whatever is pending and makes `stream` wait for every group issued so far
int blsw_engine_flush(blsw_engine_t* e, void* stream_) {
    if (!e) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    int rc = launch_group(e);
    e->ramp_pos = 0;  // the pipeline drains: the next submits start with small groups again (options.group_ramp)
    if (rc) return rc;
    if ((rc = pump(e))) return rc;
    for (int k = 0; k < e->nbuf; k++)
        if (e->buf[k].used && e->buf[k].jobs_left == 0) hipStreamWaitEvent(st, e->buf[k].ev_done, 0);
    return hip_ok(hipGetLastError(), "flush");
}

int blsw_engine_submitted(blsw_engine_t* e, uint64_t* seq) {
    if (!e || !seq) return BLSW_ERR_ARG;
    *seq = e->submitted;
    return BLSW_OK;
}
int blsw_engine_launched(blsw_engine_t* e, uint64_t* seq) {
    if (!e || !seq) return BLSW_ERR_ARG;
    *seq = e->launched;
    return BLSW_OK;
}
int blsw_engine_materialised(blsw_engine_t* e, uint64_t* seq) {
    if (!e || !seq) return BLSW_ERR_ARG;
    *seq = e->materialised;
    return BLSW_OK;
}
// Step `seq` must have been issued (seq < launched) and its group buffer not yet recycled (at most n_buffers groups back:
// older steps completed before their buffer was reused, so there is nothing to wait for).
int blsw_engine_wait_step(blsw_engine_t* e, uint64_t seq, void* stream_) {
    if (!e || seq >= e->launched) return BLSW_ERR_ARG;
    if (seq >= e->materialised) return BLSW_ERR_BUSY;  // consumer mode: its output is still held by an earlier step
    DeviceGuard guard(e->device);
    for (int k = 0; k < e->nbuf; k++) {
        GroupBuf& b = e->buf[k];
        if (b.used && seq >= b.first_seq && seq < b.first_seq + b.steps)
            return hip_ok(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream_), b.ev_step[seq - b.first_seq], 0), "wait step");
    }
    return BLSW_OK;
}
int blsw_engine_output_consumed(blsw_engine_t* e, const void* d_output, void* stream_) {
    if (!e || !d_output) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    int slot = consumed_slot(e, d_output);
    if (slot < 0) slot = free_consumed_slot(e);
    if (slot < 0) return BLSW_ERR_BUSY;  // more than BLSW_MAX_CONSUMED distinct outputs with a pending release, a held or an accepted step
    if (!e->consumed_ev[slot] && hip_ok(hipEventCreateWithFlags(&e->consumed_ev[slot], hipEventDisableTiming), "event create")) return BLSW_ERR_HIP;
    if (hip_ok(hipEventRecord(e->consumed_ev[slot], reinterpret_cast<hipStream_t>(stream_)), "event record")) return BLSW_ERR_HIP;
    e->consumed_ptr[slot] = d_output;
    e->consumed_live[slot] = true;
    e->held[slot] = false;
    return pump(e);  // consumer mode: steps that waited for this output go out now
}

// Average duration (ms) of the k_sha_expand launches issued since the last call (HIP events recorded on the stream the
// kernel ran on); blocks until they have finished, then resets the statistics. count may be 0.
int blsw_engine_expand_stats(blsw_engine_t* e, uint32_t* count, float* avg_ms) {
    if (!e || !count || !avg_ms) return BLSW_ERR_ARG;
    DeviceGuard guard(e->device);
    double sum = 0;
    for (uint32_t i = 0; i < e->n_timed; i++) {
        if (hip_ok(hipEventSynchronize(e->ev_exp[2 * i + 1]), "event sync")) return BLSW_ERR_HIP;
        float ms = 0;
        if (hip_ok(hipEventElapsedTime(&ms, e->ev_exp[2 * i], e->ev_exp[2 * i + 1]), "event elapsed")) return BLSW_ERR_HIP;
        sum += ms;
    }
    *count = e->n_timed;
    *avg_ms = e->n_timed ? (float)(sum / e->n_timed) : 0.f;
    e->n_timed = 0;
    return BLSW_OK;
}

int blsw_witness_digest(const uint64_t* d_witness, uint64_t witness_stride, uint64_t n, uint32_t n_witness, uint64_t* d_digest, void* stream_) {
    if (!d_witness || !d_digest || n == 0 || n_witness == 0 || witness_stride < n_witness) return BLSW_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    DeviceGuard guard(stream_device(st));
    if (hip_ok(hipMemsetAsync(d_digest, 0, n * 2 * sizeof(uint64_t), st), "memset")) return BLSW_ERR_HIP;
    const uint64_t n_words = (uint64_t)n_witness * 6, per_block = 2ull * 256 * BLSW_DIGEST_ITERS;
    const uint64_t chunks = (n_words + per_block - 1) / per_block;
    // one row of workgroups per instance (grid.y <= 65535): larger batches in slices of 65 535 instances
    for (uint64_t first = 0; first < n; first += 65535) {
        const uint64_t cnt = n - first < 65535 ? n - first : 65535;
        dim3 grid((unsigned)(chunks < BLSW_DIGEST_MAX_BLOCKS ? chunks : BLSW_DIGEST_MAX_BLOCKS), (unsigned)cnt);  // a workgroup walks its instance's chunks with stride grid.x
        hipLaunchKernelGGL(k_digest, grid, dim3(256), 0, st, d_witness + first * witness_stride * 6, witness_stride, n_words, d_digest + 2 * first);
    }
    return hip_ok(hipGetLastError(), "launch");
}

// one-step descriptor at the head of a caller workspace (direct-mode entry points); `h` is copied before returning
static int put_desc(StepDesc* d_desc, const StepDesc& h, hipStream_t st) {
    if (hip_ok(hipMemcpyAsync(d_desc, &h, sizeof(h), hipMemcpyHostToDevice, st), "memcpy")) return BLSW_ERR_HIP;
    return hip_ok(hipStreamSynchronize(st), "sync");  // `h` is a stack object
}
static Group direct_group(uint64_t n, uint32_t K, uint32_t msg_len, const blsw_layout_t& L, StepDesc* d_desc, const Workspace& ws) {
    Group g;
    g.N = n * K;
    g.n = (uint32_t)n;
    g.K = K;
    g.msg_len = msg_len;
    g.desc = d_desc;
    g.L = L;
    g.LS = L;
    g.ws = ws;
    g.chain_prio = 0;
    g.canonical = 0;
    return g;
}

int blsw_hash_to_g2_batch(const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint64_t* d_out_affine, void* d_workspace, uint64_t workspace_bytes,
                          void* stream_) {
    if ((!d_msg && msg_len) || n == 0 || n > 0x7fffffffu || !d_workspace || !d_out_affine || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    // the step descriptor lives at the head of the workspace
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(d_workspace);
    Workspace ws = carve(reinterpret_cast<char*>(d_workspace) + 256, n, L, false, DEFAULT_MODES);
    if (ws.total_bytes + 256 > workspace_bytes) return BLSW_ERR_WORKSPACE;
    Group g = direct_group(n, 1, msg_len, L, d_desc, ws);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    DeviceGuard guard(stream_device(st));  // the device that owns `stream`
    StepDesc h = {nullptr, nullptr, d_msg, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (int rc = put_desc(d_desc, h, st)) return rc;
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64);
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_map_values, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_h_to_affine, dim3(g1), dim3(64), 0, st, n, g.ws, d_out_affine);
    return hip_ok(hipGetLastError(), "launch");
}
// BLS::verify (bls.rs:427-458) for a batch of compressed (pk, sig) and messages, as VALUES (no circuit): decode with the endomorphism subgroup checks,
// the value-only hash to G2, projective line coefficients and a two-pair Miller loop + final exponentiation on the six-lane team (vpairing.hpp).
// d_result[i] = 1 iff both points decode to non-identity subgroup points and e(-g1, sig) e(pk, H(m)) = 1; d_status [n][2] as blsw_decode_batch.
static uint64_t verify_workspace(uint64_t n, const blsw_layout_t& L, uint64_t* off_pk, uint64_t* off_sig, uint64_t* off_ls, uint64_t* off_lh, uint64_t* off_ws) {
    uint64_t o = 256;  // the step descriptor
    *off_pk = o;
    o = align_up(o + n * 96, 256);
    *off_sig = o;
    o = align_up(o + n * 192, 256);
    *off_ls = o;
    o = align_up(o + (uint64_t)BLSW_VLINE_ROWS * n * sizeof(Fp), 256);
    *off_lh = o;
    o = align_up(o + (uint64_t)BLSW_VLINE_ROWS * n * sizeof(Fp), 256);
    *off_ws = o;
    return o + carve(nullptr, n, L, false, DEFAULT_MODES).total_bytes;
}
int blsw_verify_workspace_bytes(uint64_t n, uint32_t msg_len, uint64_t* bytes) {
    if (!bytes || n == 0 || n > 0x7fffffffu || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    uint64_t a, b, c, d, e;
    *bytes = verify_workspace(n, L, &a, &b, &c, &d, &e);
    return BLSW_OK;
}
int blsw_verify_batch(const uint8_t* d_pk48, const uint8_t* d_sig96, const uint8_t* d_msg, uint32_t msg_len, uint64_t n, int32_t* d_result, int32_t* d_status,
                      void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    if (!d_pk48 || !d_sig96 || (!d_msg && msg_len) || n == 0 || n > 0x7fffffffu || !d_result || !d_status || !d_workspace || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    uint64_t off_pk, off_sig, off_ls, off_lh, off_ws;
    if (verify_workspace(n, L, &off_pk, &off_sig, &off_ls, &off_lh, &off_ws) > workspace_bytes) return BLSW_ERR_WORKSPACE;
    char* base = reinterpret_cast<char*>(d_workspace);
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(base);
    uint64_t* pk_xy = reinterpret_cast<uint64_t*>(base + off_pk);
    uint64_t* sig_xy = reinterpret_cast<uint64_t*>(base + off_sig);
    Group g = direct_group(n, 1, msg_len, L, d_desc, carve(base + off_ws, n, L, false, DEFAULT_MODES));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    DeviceGuard guard(stream_device(st));
    StepDesc h = {nullptr, nullptr, d_msg, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (int rc = put_desc(d_desc, h, st)) return rc;
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64);
    hipLaunchKernelGGL(k_decode, dim3(g2), dim3(64), 0, st, d_pk48, d_sig96, n, pk_xy, sig_xy, d_status);
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_map_values, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor_values, dim3(g1), dim3(64), 0, st, g);
    launch_verify_values(n, g.ws, pk_xy, sig_xy, reinterpret_cast<Fp*>(base + off_ls), reinterpret_cast<Fp*>(base + off_lh), d_status, d_result, st);
    return hip_ok(hipGetLastError(), "launch");
}
// BLS::sign + PublicKey::from(&sk) for a batch (bls.rs:411-425, 183-195). Workspace: blsw_hash_to_g2_workspace_bytes.
int blsw_sign_batch(const uint8_t* d_sk32_le, const uint8_t* d_msg, uint32_t msg_len, uint64_t n, uint8_t* d_sig96, uint64_t* d_sig_xy, uint8_t* d_pk48,
                    uint64_t* d_pk_xy, int32_t* d_status, void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    if (!d_sk32_le || (!d_msg && msg_len) || n == 0 || n > 0x7fffffffu || !d_workspace || !d_status || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(d_workspace);
    Workspace ws = carve(reinterpret_cast<char*>(d_workspace) + 256, n, L, false, DEFAULT_MODES);
    if (ws.total_bytes + 256 > workspace_bytes) return BLSW_ERR_WORKSPACE;
    Group g = direct_group(n, 1, msg_len, L, d_desc, ws);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    DeviceGuard guard(stream_device(st));  // the device that owns `stream`
    StepDesc h = {nullptr, nullptr, d_msg, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (int rc = put_desc(d_desc, h, st)) return rc;
    const unsigned g1 = (unsigned)((n + 63) / 64), g2 = (unsigned)((2 * n + 63) / 64);
    hipLaunchKernelGGL(k_sha_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_map_values, dim3(g2), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_cofactor_values, dim3(g1), dim3(64), 0, st, g);
    hipLaunchKernelGGL(k_sign, dim3(g2), dim3(64), 0, st, n, g.ws, d_sk32_le, d_sig96, d_sig_xy, d_pk48, d_pk_xy, d_status);
    return hip_ok(hipGetLastError(), "launch");
}
int blsw_layout_aggregate(uint32_t msg_len, uint32_t n_keys, blsw_layout_t* out) {
    if (!out || msg_len > 65535) return BLSW_ERR_ARG;
    make_layout(msg_len, out, n_keys);
    return BLSW_OK;
}
static uint64_t agg_workspace(uint64_t n, const blsw_layout_t& L, uint64_t* off_desc, uint64_t* off_keyproj, uint64_t* off_ws) {
    uint64_t o = 0;
    *off_desc = o;
    o = align_up(o + sizeof(StepDesc), 256);
    *off_keyproj = o;
    o = align_up(o + 3ull * n * L.n_keys * sizeof(Fp), 256);
    *off_ws = o;
    return o + carve(nullptr, n, L, false, DEFAULT_MODES).total_bytes;
}
int blsw_aggregate_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t n_keys, uint64_t* bytes) {
    if (!bytes || n == 0 || n_keys == 0 || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L, n_keys);
    uint64_t a, b, c;
    *bytes = agg_workspace(n, L, &a, &b, &c);
    return BLSW_OK;
}
int blsw_aggregate_verify_batch(const uint64_t* d_pks_xy, const uint8_t* d_bitmap, uint32_t n_keys, const uint64_t* d_sig_xy, const uint8_t* d_msg,
                                uint32_t msg_len, uint64_t n, uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, uint32_t* d_count,
                                void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    if (!d_pks_xy || !d_bitmap || n_keys == 0 || !d_sig_xy || (!d_msg && msg_len) || n == 0 || n > 65535 || !d_workspace || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L, n_keys);
    if (d_witness && witness_stride < L.n_witness) return BLSW_ERR_ARG;
    uint64_t off_desc, off_keyproj, off_ws;
    if (agg_workspace(n, L, &off_desc, &off_keyproj, &off_ws) > workspace_bytes) return BLSW_ERR_WORKSPACE;
    char* base = reinterpret_cast<char*>(d_workspace);
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(base + off_desc);
    Fp* keyproj = reinterpret_cast<Fp*>(base + off_keyproj);
    Group g = direct_group(n, 1, msg_len, L, d_desc, carve(base + off_ws, n, L, false, DEFAULT_MODES));
    const ChainKernels ck = chain_kernels(true);  // direct mode: few waves, latency-bound
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    DeviceGuard guard(stream_device(st));  // the device that owns `stream`
    StepDesc h = {nullptr, d_sig_xy, d_msg, d_witness, witness_stride, d_result, d_pks_xy, d_bitmap, d_count};
    if (int rc = put_desc(d_desc, h, st)) return rc;
    const unsigned g1 = (unsigned)((n + 63) / 64), gk = (unsigned)((n * n_keys + 63) / 64);
    hipLaunchKernelGGL(ck.agg_keys, dim3(gk), dim3(64), 0, st, g, keyproj);
    hipLaunchKernelGGL(ck.agg_sum, dim3(g1), dim3(64), 0, st, g, (const Fp*)keyproj);
    const Latency lat = {g.ws.cofv != nullptr, g.ws.cofv != nullptr};  // a small direct call is latency-bound: quads, values-first cofactor chain
    launch_g2_alloc(ck, lat, g, st);
    launch_prepare(ck, lat, g, 1, st);
    hipLaunchKernelGGL(ck.sha, dim3(g1), dim3(64), 0, st, g, d_witness ? 1 : 0, 1);
    if (d_witness) {
        ExpandArgs xa = {g.ws.bits, g.ws.sha_words, 0, g.L.sha_bits, g.L.off_expand, d_witness, witness_stride, 1u, 0u, 0};
        launch_expand(BLSW_DEFAULT_EXPAND_VARIANT, 0, 0, st, xa, (unsigned)n);
    }
    launch_map(ck, lat, g, st);
    launch_cofactor_direct(ck, lat, g, st, stream_device(st));
    launch_prepare(ck, lat, g, 0, st);
    launch_pairing(g, DEFAULT_MODES, st);
    return hip_ok(hipGetLastError(), "launch");
}

// ---- N+1-pair product of pairings (one signature over n_pairs (pk, msg) pairs)
int blsw_layout_multi(uint32_t msg_len, uint32_t n_pairs, blsw_layout_t* out) {
    if (!out || msg_len > 65535 || n_pairs == 0 || n_pairs > 4096) return BLSW_ERR_ARG;
    make_layout(msg_len, out, 0, n_pairs);
    // the witness vector must stay addressable with 32-bit element offsets
    const uint64_t total = (uint64_t)out->off_prep_h - out->off_expand;  // n_pairs * stride_hash
    if (total / n_pairs != out->stride_hash) return BLSW_ERR_ARG;
    return BLSW_OK;
}
int blsw_verify_multi_workspace_bytes(uint64_t n, uint32_t msg_len, uint32_t n_pairs, uint64_t* bytes) {
    blsw_layout_t L;
    if (!bytes || n == 0 || blsw_layout_multi(msg_len, n_pairs, &L)) return BLSW_ERR_ARG;
    *bytes = 256 + carve(nullptr, n * n_pairs, L, false, DEFAULT_MODES, n).total_bytes + miller_par_bytes(n, n_pairs, BLSW_MILLER_CHUNK_DEFAULT);
    return BLSW_OK;
}
int blsw_verify_multi_batch(const uint64_t* d_pks_xy, const uint8_t* d_msgs, uint32_t msg_len, uint32_t n_pairs, const uint64_t* d_sig_xy, uint64_t n,
                            uint64_t* d_witness, uint64_t witness_stride, int32_t* d_result, void* d_workspace, uint64_t workspace_bytes, void* stream_) {
    blsw_layout_t L;
    if (!d_pks_xy || (!d_msgs && msg_len) || !d_sig_xy || n == 0 || !d_workspace || blsw_layout_multi(msg_len, n_pairs, &L)) return BLSW_ERR_ARG;
    const uint64_t NP = n * n_pairs;  // per-pair lanes
    if (NP > 65535 * 16ull || n > 65535) return BLSW_ERR_ARG;
    if (d_witness && witness_stride < L.n_witness) return BLSW_ERR_ARG;
    StepDesc* d_desc = reinterpret_cast<StepDesc*>(d_workspace);
    Workspace ws = carve(reinterpret_cast<char*>(d_workspace) + 256, NP, L, false, DEFAULT_MODES, n);
    const bool par = n_pairs >= BLSW_MILLER_PAR_MIN_PAIRS;  // pairs in parallel (miller_par.hpp); few pairs: one team walks the chain
    if (ws.total_bytes + 256 + (par ? miller_par_bytes(n, n_pairs, BLSW_MILLER_CHUNK_DEFAULT) : 0) > workspace_bytes) return BLSW_ERR_WORKSPACE;
    Group gp = direct_group(n, n_pairs, msg_len, L, d_desc, ws);  // per-pair work: N = n * n_pairs lanes
    Group gs = direct_group(n, 1, msg_len, L, d_desc, ws);        // per-signature work: N = n lanes
    const ChainKernels ck = chain_kernels(true);                  // direct mode: few waves, latency-bound
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    const int dev = stream_device(st);  // the device that owns `stream`
    DeviceGuard guard(dev);
    StepDesc h = {d_pks_xy, d_sig_xy, d_msgs, d_witness, witness_stride, d_result, nullptr, nullptr, nullptr};
    if (int rc = put_desc(d_desc, h, st)) return rc;
    const unsigned p1 = (unsigned)((NP + 63) / 64);
    // fork: the signature's allocation + prepare (one lane per instance: 57 ms of latency) and the keys' allocation run beside the
    // hash-to-G2 chains of the pairs; join in front of the Miller product. The two side streams and three events are created once
    // per host thread and device and kept (an event is re-recorded per call; a wait refers to the record that preceded it).
    struct Side {
        int device = -1;
        hipStream_t aux[3] = {nullptr, nullptr, nullptr};
        hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
        bool ok = false;
    };
    static thread_local std::map<int, Side> sides;  // per host thread and device ordinal (the device that owns `stream`)
    Side& sd = sides[dev];
    if (sd.device != dev) {
        sd.device = dev;
        sd.ok = hipStreamCreateWithFlags(&sd.aux[0], hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&sd.aux[1], hipStreamNonBlocking) == hipSuccess &&
                hipStreamCreateWithFlags(&sd.aux[2], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&sd.ev_fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&sd.ev_join[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&sd.ev_join[1], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&sd.ev_join[2], hipEventDisableTiming) == hipSuccess;
    }
    const bool forked = sd.ok;
    hipEvent_t ev_fork = sd.ev_fork;
    hipEvent_t* ev_join = sd.ev_join;
    hipStream_t s_sig = forked ? sd.aux[0] : st, s_keys = forked ? sd.aux[1] : st;
    if (forked) {
        hipEventRecord(ev_fork, st);  // the descriptor copy
        hipStreamWaitEvent(s_sig, ev_fork, 0);
        hipStreamWaitEvent(s_keys, ev_fork, 0);
    }
    const Latency lat = {ws.cofv != nullptr, ws.cofv != nullptr};  // a small direct call is latency-bound: quads, values-first cofactor chain
    launch_g2_alloc(ck, lat, gs, s_sig);
    launch_prepare(ck, lat, gs, 1, s_sig);
    hipLaunchKernelGGL(ck.g1, dim3(p1), dim3(64), 0, s_keys, gp);
    // the SHA witness bits and their expansion (92 % of the output bytes) need only the messages: their own stream, beside the curve
    // chains; the chains start from the value-only hash_to_field
    hipStream_t s_exp = forked ? sd.aux[2] : st;
    if (forked) hipStreamWaitEvent(s_exp, ev_fork, 0);
    if (d_witness) {
        hipLaunchKernelGGL(ck.sha, dim3(p1), dim3(64), 0, s_exp, gp, 1, 0);
        // blockIdx.y = flat (instance, pair); grid.y <= 65535: several launches for larger batches
        const uint64_t per_launch = (65535 / n_pairs) * (uint64_t)n_pairs;
        for (uint64_t first = 0; first < NP; first += per_launch) {
            const uint64_t cnt = NP - first < per_launch ? NP - first : per_launch;
            ExpandArgs xa = {ws.bits, ws.sha_words, first, L.sha_bits, L.off_expand, d_witness + (first / n_pairs) * witness_stride * 6, witness_stride, n_pairs, L.stride_hash, 0};
            launch_expand(BLSW_DEFAULT_EXPAND_VARIANT, 0, 0, s_exp, xa, (unsigned)cnt);
        }
    }
    hipLaunchKernelGGL(k_sha_values, dim3(p1), dim3(64), 0, st, gp);
    launch_map(ck, lat, gp, st);
    launch_cofactor_direct(ck, lat, gp, st, dev);
    launch_prepare(ck, lat, gp, 0, st);
    if (forked) {
        hipEventRecord(ev_join[0], s_sig);
        hipEventRecord(ev_join[1], s_keys);
        hipEventRecord(ev_join[2], s_exp);
        hipStreamWaitEvent(st, ev_join[0], 0);
        hipStreamWaitEvent(st, ev_join[1], 0);
        hipStreamWaitEvent(st, ev_join[2], 0);
    }
    if (!par) {
        hipLaunchKernelGGL(k_pairing_team_multi, dim3((unsigned)((n + BLSW_TEAMS_PER_WAVE - 1) / BLSW_TEAMS_PER_WAVE)), dim3(64), 0, st, gs, n_pairs, NP);
        return hip_ok(hipGetLastError(), "launch");
    }
    MillerParArgs ma;
    ma.K = n_pairs;
    ma.B = BLSW_MILLER_CHUNK_DEFAULT;
    ma.C = (n_pairs + ma.B - 1) / ma.B;
    ma.n_h = NP;
    ma.spine_lane = 0;
    {
        char* p = reinterpret_cast<char*>(d_workspace) + align_up(256 + ws.total_bytes, 256);
        auto take = [&](uint64_t items) {
            Fp* r = reinterpret_cast<Fp*>(p);
            p += align_up(items * 12 * sizeof(Fp), 256);
            return r;
        };
        ma.cprod = take(n * 68 * ma.C);
        ma.q = take(n * 68 * ma.C);
        ma.t = take(n * 68);
        ma.f1 = take(n * 68);
        ma.ffinal = take(n);
    }
    launch_miller_par(gs, ma, st, forked ? s_sig : nullptr, ev_fork, ev_join[0]);
    return hip_ok(hipGetLastError(), "launch");
}

int blsw_decode_batch(const uint8_t* d_pk48, const uint8_t* d_sig96, uint64_t n, uint64_t* d_pk_xy, uint64_t* d_sig_xy, int32_t* d_status, void* stream_) {
    if (!d_pk48 || !d_sig96 || !d_pk_xy || !d_sig_xy || !d_status || n == 0 || n > 0x3fffffffu) return BLSW_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    DeviceGuard guard(stream_device(st));  // the device that owns `stream`
    hipLaunchKernelGGL(k_decode, dim3((unsigned)((2 * n + 63) / 64)), dim3(64), 0, st, d_pk48, d_sig96, n, d_pk_xy, d_sig_xy, d_status);
    return hip_ok(hipGetLastError(), "launch");
}
// Signature::aggregate / PublicKey::aggregate for n lists of k compressed points (bls.rs:288-300, 183-195)
int blsw_aggregate_points_workspace_bytes(uint32_t group, uint64_t n, uint32_t k, uint64_t* bytes) {
    if (!bytes || (group != 1 && group != 2) || n == 0 || k == 0 || n > 0x3fffffffu / k) return BLSW_ERR_ARG;
    const uint64_t m = n * k;
    *bytes = align_up(m * (group == 1 ? 2 : 4) * sizeof(Fp), 256) + align_up(m * sizeof(int32_t), 256);
    return BLSW_OK;
}
int blsw_aggregate_points_batch(uint32_t group, const uint8_t* d_in, uint32_t k, uint64_t n, uint8_t* d_out, int32_t* d_status, void* d_workspace,
                                uint64_t workspace_bytes, void* stream_) {
    uint64_t need = 0;
    if (!d_in || !d_out || !d_status || !d_workspace || blsw_aggregate_points_workspace_bytes(group, n, k, &need)) return BLSW_ERR_ARG;
    if (workspace_bytes < need) return BLSW_ERR_WORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    DeviceGuard guard(stream_device(st));
    const uint64_t m = n * k;
    Fp* xy = reinterpret_cast<Fp*>(d_workspace);
    int32_t* pst = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(d_workspace) + align_up(m * (group == 1 ? 2 : 4) * sizeof(Fp), 256));
    hipLaunchKernelGGL(k_decode_points, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, group, d_in, m, xy, pst);
    hipLaunchKernelGGL(k_sum_points, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, group, (const Fp*)xy, (const int32_t*)pst, k, n, d_out, d_status);
    return hip_ok(hipGetLastError(), "launch");
}
int blsw_hash_to_g2_workspace_bytes(uint64_t n, uint32_t msg_len, uint64_t* bytes) {
    if (!bytes || n == 0 || msg_len > 65535) return BLSW_ERR_ARG;
    blsw_layout_t L;
    make_layout(msg_len, &L);
    *bytes = carve(nullptr, n, L, false, DEFAULT_MODES).total_bytes + 256;
    return BLSW_OK;
}

// Plain fill of a caller buffer (e.g. a witness tensor before it is used) in the expansion's store geometry, `reps` times after one warm-up pass, on the
// NULL stream of the current device; bytes per second of the timed passes. Synchronous. The buffer's contents are overwritten.
int blsw_fill_rate(void* d_buf, uint64_t bytes, uint32_t reps, double* bytes_per_s) {
    if (!d_buf || bytes < 16 || reps == 0 || !bytes_per_s || (reinterpret_cast<uintptr_t>(d_buf) & 15)) return BLSW_ERR_ARG;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = hip_ok(hipEventCreate(&e0), "event create");
    if (!rc) rc = hip_ok(hipEventCreate(&e1), "event create");
    const uint64_t n16 = bytes / 16;
    const unsigned grid = (unsigned)((n16 + 3071) / 3072);
    if (!rc) {
        hipLaunchKernelGGL(k_bench_fill, dim3(grid), dim3(384), 0, 0, reinterpret_cast<uint4*>(d_buf), n16);
        hipEventRecord(e0, 0);
        for (uint32_t r = 0; r < reps; r++) hipLaunchKernelGGL(k_bench_fill, dim3(grid), dim3(384), 0, 0, reinterpret_cast<uint4*>(d_buf), n16);
        hipEventRecord(e1, 0);
        rc = hip_ok(hipEventSynchronize(e1), "event sync");
    }
    if (!rc) {
        float ms = 0;
        rc = hip_ok(hipEventElapsedTime(&ms, e0, e1), "event elapsed");
        if (!rc) *bytes_per_s = (double)n16 * 16.0 * reps / (ms * 1e-3);
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    return rc;
}

// which = 0: v_mad_u64_u32 issue rate (multiply-adds/s); 1: fp_mul (Fp products/s); 2: fp_inv (inversions/s); 3: Fp products/s inside
// witness-emitting Fp2 mul + sqr. Synchronous, on the current device.
int blsw_microbench(int which, uint32_t iters, uint32_t blocks, double* ops_per_s) {
    if (!ops_per_s || iters == 0 || blocks == 0 || which < 0 || which > 4) return BLSW_ERR_ARG;
    uint32_t* d = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = hip_ok(hipMalloc(&d, 4), "malloc");
    if (!rc) rc = hip_ok(hipEventCreate(&e0), "event create");
    if (!rc) rc = hip_ok(hipEventCreate(&e1), "event create");
    const int threads = which == 0 ? 256 : 64;
    const double per_iter[5] = {8.0, 2.0, 1.0, 5.0, 2.0};  // MADs, fp products, fp inversions, fp products (one Fp2 mul + one Fp2 sqr), fp products (32-bit CIOS)
    for (int rep = 0; rep < 2 && !rc; rep++) {        // first pass warms up
        hipEventRecord(e0, 0);
        if (which == 0)
            hipLaunchKernelGGL(k_bench_mad, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else if (which == 1)
            hipLaunchKernelGGL(k_bench_fpmul, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else if (which == 2)
            hipLaunchKernelGGL(k_bench_fpinv, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else if (which == 3)
            hipLaunchKernelGGL(k_bench_fp2mulw, dim3(blocks), dim3(threads), 0, 0, iters, d);
        else
            hipLaunchKernelGGL(k_bench_fpmul32, dim3(blocks), dim3(threads), 0, 0, iters, d);
        hipEventRecord(e1, 0);
        rc = hip_ok(hipEventSynchronize(e1), "event sync");
    }
    if (!rc) {
        float ms = 0;
        rc = hip_ok(hipEventElapsedTime(&ms, e0, e1), "event elapsed");
        if (!rc) *ops_per_s = per_iter[which] * iters * blocks * threads / (ms * 1e-3);
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (d) hipFree(d);
    return rc;
}
}
