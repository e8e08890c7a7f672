// One host thread driving a row partition: a GROUP context is a head handle over P ordinary contexts
// ("ranks", one worker thread each), so that a serial caller - the reference's Fortran host is one - sees
// ONE context with whole vectors and whole reference-layout arrays while the generator rows, the Krylov
// basis and w are partitioned over P devices (RCCL between them), or over P contexts of one device through
// the loop-back transport (one-GPU rehearsal).  Every entry point of include/kfsp.h that gets a head fans
// out to the ranks through their PUBLIC entry points - the same calls a one-process-per-GPU launcher makes
// on its own rank - and brings the results back; scalars that every rank must agree on (beta, H, AVNORM,
// WSUM, the drop plan) are compared bit for bit on the way (4002 if they ever differ).
//
// kfsp_dgexpv / kfsp_dgexpv_replay / kfsp_expv_fixed are clients of this ABI: on a head they run once, on
// the caller's thread, and every device call they make fans out.  The drop / expand callbacks therefore run
// once as well, on the caller's one copy of the state space (KrylovSolver.f90:509-534), and upload through
// the head.
#include "kfsp_ctx.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>

namespace kfsp {

struct Group {
    int n = 0;
    std::vector<kfsp_ctx *> sub;
    void *loop = nullptr;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    std::function<int(int)> job;
    std::vector<int> rc;
    std::vector<char> done;
    // Watchdog.  A rank that fails BEFORE a collective leaves its peers inside that collective for ever (RCCL), or until
    // the loop-back barrier's own 120 s guard; a rank that hangs keeps everyone.  So run() waits with two clocks: once
    // any rank has returned an error the others get grace_s to come back, and the whole fan-out gets timeout_s; when
    // either expires the communicators of ALL ranks are aborted (comm_abort: ncclCommAbort / the loop-back release), which
    // makes the blocked ranks return, the failing rank's error is reported ("rank p: ..."), and the group is BROKEN:
    // every later call returns 2999 at once (a partition whose ranks have lost step cannot be continued - destroy it and
    // make a new one).  Ranks that do not return even after the abort (settle_s) are left behind: the group is STUCK, its
    // threads are detached at destruction, code 2998.  Nothing here ever ends the process.
    // Options on the head: "group_timeout_ms", "group_grace_ms", "group_settle_ms"; environment KFSP_GROUP_TIMEOUT_S,
    // KFSP_GROUP_GRACE_S for callers without access to options (the Fortran host).
    double timeout_s = 1800.0, grace_s = 15.0, settle_s = 30.0;
    bool broken = false, stuck = false;
    int inject_rank = -1;              // test hook (option "group_inject_failure"): this rank fails its next job with -77
    std::function<void()> abort_all;   // releases ranks blocked in a collective

    void worker(int p)
    {
        uint64_t seen = 0;
        for (;;) {
            std::function<int(int)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
                f = job;
            }
            int r = 4000;
            try {
                r = f(p);
            } catch (...) {
                r = 4000;
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                rc[(size_t)p] = r;
                done[(size_t)p] = 1;
                --pending;
                if (pending == 0 || r != 0) cv_done.notify_all();
            }
        }
    }

    // f(p) on the thread of rank p, all ranks at once; first non-zero return code (and that rank).  *who = -1 with 2999 /
    // 2998: the group was aborted earlier, or no rank failed and the deadline expired.
    int run(const std::function<int(int)> &f, int *who = nullptr)
    {
        using clock = std::chrono::steady_clock;
        if (who) *who = -1;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (stuck) return 2998;
            if (broken) return 2999;
            job = f;
            pending = n;
            std::fill(done.begin(), done.end(), 0);
            std::fill(rc.begin(), rc.end(), 0);
            ++gen;
        }
        cv_go.notify_all();
        bool expired = false;
        {
            std::unique_lock<std::mutex> lk(mu);
            const clock::time_point t_end = clock::now() + std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(timeout_s));
            clock::time_point t_fail = clock::time_point::max();
            while (pending != 0) {
                if (t_fail == clock::time_point::max())
                    for (int p = 0; p < n; ++p)
                        if (done[(size_t)p] && rc[(size_t)p] != 0)
                            t_fail = clock::now() + std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(grace_s));
                const clock::time_point wake = std::min(t_end, t_fail);
                if (clock::now() >= wake) {
                    expired = true;
                    break;
                }
                cv_done.wait_until(lk, wake);
            }
        }
        if (expired) {
            if (abort_all) abort_all();
            std::unique_lock<std::mutex> lk(mu);
            broken = true;
            const bool back = cv_done.wait_for(lk, std::chrono::duration<double>(settle_s), [&] { return pending == 0; });
            if (!back) stuck = true;
        }
        std::lock_guard<std::mutex> lk(mu);
        // the rank that failed on its own is the one to name: ranks released by the abort report 2999
        for (int p = 0; p < n; ++p)
            if (done[(size_t)p] && rc[(size_t)p] != 0 && !(expired && rc[(size_t)p] == 2999)) {
                if (who) *who = p;
                return rc[(size_t)p];
            }
        if (expired) return stuck ? 2998 : 2999;
        return 0;
    }
};

namespace {

int gfail(kfsp_ctx *h, int code, const char *what)
{
    h->err = what;
    return code;
}

// run on all ranks; on failure the head's error text is the failing rank's
int gall(kfsp_ctx *h, const std::function<int(kfsp_ctx *, int)> &f)
{
    Group *g = h->group;
    int who = -1;
    const int inject = g->inject_rank;
    g->inject_rank = -1;
    const int rc = g->run([&](int p) {
        if (p == inject) {
            g->sub[(size_t)p]->err = "injected failure (option group_inject_failure)";
            return -77;
        }
        return f(g->sub[(size_t)p], p);
    }, &who);
    if (rc) {
        if (who >= 0) h->err = std::string("rank ") + std::to_string(who) + ": " + kfsp_last_error(g->sub[(size_t)who]);
        else h->err = rc == 2998 ? "group context: ranks did not return even after their communicators were aborted"
                                 : "group context: aborted (a rank failed or the deadline of a call expired earlier); destroy it";
        if (g->broken && who >= 0) h->err += " [the other ranks were released by aborting the communicator; the group is unusable now]";
    }
    return rc;
}

struct Block {
    int64_t row0, nrows;
};
Block block_of(const kfsp_ctx *h, int64_t n, int p)
{
    Block b{0, 0};
    (void)kfsp_partition(n, h->group->n, p, &b.row0, &b.nrows, nullptr);
    return b;
}

bool same_bits(const double *a, const double *b, size_t k) { return std::memcmp(a, b, k * sizeof(double)) == 0; }

}  // namespace

// ------------------------------------------------------------------------------------------------
int group_create(int nranks, const int *devices, kfsp_ctx **out)
{
    if (nranks < 1 || nranks > 64) return -1;
    if (!out) return -3;
    *out = nullptr;
    std::unique_ptr<kfsp_ctx> head(new (std::nothrow) kfsp_ctx);
    std::unique_ptr<Group> g(new (std::nothrow) Group);
    if (!head || !g) return 4001;
    g->n = nranks;
    g->sub.assign((size_t)nranks, nullptr);
    g->rc.assign((size_t)nranks, 0);
    g->done.assign((size_t)nranks, 0);
    if (const char *e = std::getenv("KFSP_GROUP_TIMEOUT_S")) {
        const double v = std::atof(e);
        if (v > 0.0) g->timeout_s = v;
    }
    if (const char *e = std::getenv("KFSP_GROUP_GRACE_S")) {
        const double v = std::atof(e);
        if (v > 0.0) g->grace_s = v;
    }
    // distinct devices: RCCL between them; a device named twice: the loop-back transport
    bool distinct = true;
    for (int p = 0; p < nranks; ++p)
        for (int q = 0; q < p; ++q)
            if ((devices ? devices[p] : 0) == (devices ? devices[q] : 0)) distinct = false;
    if (!devices && nranks > 1) distinct = false;
    int rc = 0;
    for (int p = 0; p < nranks && !rc; ++p) rc = kfsp_create(devices ? devices[p] : 0, &g->sub[(size_t)p]);
    if (!rc && !distinct && nranks > 1) rc = kfsp_loopback_create(nranks, &g->loop);
    if (rc) {
        for (kfsp_ctx *c : g->sub)
            if (c) (void)kfsp_destroy(c);
        return rc;
    }
    head->device = devices ? devices[0] : 0;
    head->group = g.get();
    try {
        g->th.reserve((size_t)nranks);
        for (int p = 0; p < nranks; ++p) g->th.emplace_back(&Group::worker, g.get(), p);
    } catch (...) {
        // (a thread could not be started: stop the ones that were, or their destructors would terminate the process)
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->quit = true;
        }
        g->cv_go.notify_all();
        for (std::thread &t : g->th) t.join();
        for (kfsp_ctx *c : g->sub) (void)kfsp_destroy(c);
        if (g->loop) (void)kfsp_loopback_destroy(g->loop);
        return 4001;
    }
    kfsp_ctx *h = head.get();
    {
        Group *gp = g.get();
        gp->abort_all = [gp] {
            for (kfsp_ctx *c : gp->sub)
                if (c) comm_abort(c);
        };
    }
    if (nranks > 1) {
        unsigned char id[KFSP_UNIQUE_ID_BYTES];
        if (distinct) rc = kfsp_comm_unique_id(id);
        if (!rc)
            rc = gall(h, [&](kfsp_ctx *c, int p) {
                return distinct ? kfsp_comm_init(c, nranks, p, id) : kfsp_comm_init_loopback(c, g->loop, p);
            });
    }
    if (rc) {
        Group *gp = g.release();
        kfsp_ctx *hp = head.release();
        (void)group_destroy(hp);
        (void)gp;
        return rc;
    }
    g.release();
    *out = head.release();
    return 0;
}

int group_destroy(kfsp_ctx *h)
{
    Group *g = h->group;
    bool stuck;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        stuck = g->stuck;
        g->broken = false;             // (the contexts of a broken group are still destroyed on their own threads)
    }
    if (stuck) {
        // some rank never came back from its last call: its thread, its context and the shared state it may still touch
        // are abandoned (leaked) rather than joined - the caller gets its thread back
        for (std::thread &t : g->th) t.detach();
        delete h;
        return 2998;
    }
    (void)g->run([&](int p) { return kfsp_destroy(g->sub[(size_t)p]); });
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
        stuck = g->stuck;
    }
    if (stuck) {
        for (std::thread &t : g->th) t.detach();
        delete h;
        return 2998;
    }
    g->cv_go.notify_all();
    for (std::thread &t : g->th) t.join();
    if (g->loop) (void)kfsp_loopback_destroy(g->loop);
    delete g;
    delete h;
    return 0;
}

int group_size(const kfsp_ctx *h) { return h->group->n; }

int group_set_option(kfsp_ctx *h, const char *name, int64_t value)
{
    Group *g = h->group;
    const std::string k(name ? name : "");
    if (k == "group_timeout_ms" || k == "group_grace_ms" || k == "group_settle_ms") {
        if (value < 1) return gfail(h, -3, "a positive number of milliseconds");
        (k == "group_timeout_ms" ? g->timeout_s : k == "group_grace_ms" ? g->grace_s : g->settle_s) = 1e-3 * (double)value;
        return 0;
    }
    if (k == "group_inject_failure") {      // test hook: rank `value` fails the NEXT fan-out with -77 before doing anything
        if (value < -1 || value >= g->n) return gfail(h, -3, "no such rank");
        g->inject_rank = (int)value;
        return 0;
    }
    return gall(h, [&](kfsp_ctx *c, int) { return kfsp_set_option(c, name, value); });
}

// after any generator call: the head answers size questions like a one-rank context
static int after_matrix(kfsp_ctx *h, int rc, int64_t n)
{
    if (rc) return rc;
    h->n = n;
    h->ldv = 1;        // "a generator is set" for the argument checks the clients of this ABI share with plain contexts
    return 0;
}

int group_update_matrix_ell(kfsp_ctx *h, int32_t n, int32_t bw, int32_t ld, const int32_t *adj, const double *offdiag,
                            const double *diag, int32_t n_unchanged)
{
    return after_matrix(h, gall(h, [&](kfsp_ctx *c, int) { return kfsp_update_matrix_ell(c, n, bw, ld, adj, offdiag, diag, n_unchanged); }), n);
}

int group_set_matrix_csr(kfsp_ctx *h, int64_t n, int64_t row0, int64_t nrows, const int64_t *rowptr, const int32_t *col,
                         const double *val)
{
    if (row0 != 0 || nrows != n) return gfail(h, -3, "a group context takes the whole generator (row0 = 0, nrows = n)");
    if (!rowptr) return gfail(h, -5, "null rowptr");
    return after_matrix(h, gall(h, [&](kfsp_ctx *c, int p) {
        const Block b = block_of(h, n, p);
        std::vector<int64_t> rp((size_t)b.nrows + 1);
        const int64_t base = rowptr[b.row0];
        for (int64_t r = 0; r <= b.nrows; ++r) rp[(size_t)r] = rowptr[b.row0 + r] - base;
        return kfsp_set_matrix_csr(c, n, b.row0, b.nrows, rp.data(), col ? col + base : col, val ? val + base : val);
    }), n);
}

int group_set_matrix_box(kfsp_ctx *h, int32_t ns, const int32_t *dims, int32_t nr, const int32_t *stoich, const int32_t *ndep,
                         const int32_t *dep_species, const double *tables)
{
    int rc = gall(h, [&](kfsp_ctx *c, int) { return kfsp_set_matrix_box(c, ns, dims, nr, stoich, ndep, dep_species, tables); });
    int64_t n = 0;
    if (!rc) rc = kfsp_num_states(h->group->sub[0], &n);
    return after_matrix(h, rc, n);
}

int group_set_state_coords(kfsp_ctx *h, int32_t n, int32_t ns, int32_t ld, const int32_t *state)
{
    return gall(h, [&](kfsp_ctx *c, int) { return kfsp_set_state_coords(c, n, ns, ld, state); });
}

int group_update_state_coords(kfsp_ctx *h, int32_t n, int32_t ns, int32_t ld, const int32_t *state, int32_t n_unchanged)
{
    return gall(h, [&](kfsp_ctx *c, int) { return kfsp_update_state_coords(c, n, ns, ld, state, n_unchanged); });
}

int group_state_order_active(const kfsp_ctx *h, int *active) { return kfsp_state_order_active(h->group->sub[0], active); }

int group_matrix_info(const kfsp_ctx *h, int64_t *nrows, int64_t *slots, int64_t *nnz)
{
    int64_t a = 0, b = 0, c = 0;
    for (kfsp_ctx *s : h->group->sub) {
        int64_t x = 0, y = 0, z = 0;
        if (int rc = kfsp_matrix_info(s, &x, &y, &z)) return rc;
        a += x;
        b += y;
        c += z;
    }
    if (nrows) *nrows = a;
    if (slots) *slots = b;
    if (nnz) *nnz = c;
    return 0;
}

int group_matrix_bytes(const kfsp_ctx *h, int force_sell, int64_t *bytes)
{
    int64_t t = 0;
    for (kfsp_ctx *s : h->group->sub) {
        int64_t b = 0;
        if (int rc = kfsp_matrix_bytes(s, force_sell, &b)) return rc;
        t += b;
    }
    *bytes = t;
    return 0;
}

int group_set_vector(kfsp_ctx *h, int64_t n, const double *w)
{
    if (n != h->n) return gfail(h, -2, "n is not the size of the FSP");
    return gall(h, [&](kfsp_ctx *c, int p) {
        const Block b = block_of(h, n, p);
        return kfsp_set_vector(c, b.nrows, w ? w + b.row0 : w);
    });
}

int group_get_vector(kfsp_ctx *h, int64_t n, double *w)
{
    if (n != h->n) return gfail(h, -2, "n is not the size of the FSP");
    return gall(h, [&](kfsp_ctx *c, int p) {
        const Block b = block_of(h, n, p);
        return kfsp_get_vector(c, b.nrows, w ? w + b.row0 : w);
    });
}

int group_begin_step(kfsp_ctx *h, double *beta)
{
    Group *g = h->group;
    std::vector<double> b((size_t)g->n, 0.0);
    if (int rc = gall(h, [&](kfsp_ctx *c, int p) { return kfsp_begin_step(c, &b[(size_t)p]); })) return rc;
    for (int p = 1; p < g->n; ++p)
        if (!same_bits(&b[0], &b[(size_t)p], 1)) return gfail(h, 4002, "ranks disagree on beta");
    *beta = b[0];
    return 0;
}

int group_arnoldi(kfsp_ctx *h, int m, int jold, int qiop, double break_tol, double *H, int ldh, int *mbrkdwn, int *k1, double *avnorm)
{
    Group *g = h->group;
    if (!H || ldh < m + 2 || m < 1) return gfail(h, -6, "bad H / ldh / m");
    const size_t hsz = (size_t)ldh * (size_t)(m + 2);
    std::vector<std::vector<double>> Hp((size_t)g->n, std::vector<double>(H, H + hsz));   // columns a pass does not write are left alone
    std::vector<int> mb((size_t)g->n, 0), kk((size_t)g->n, 0);
    std::vector<double> av((size_t)g->n, 0.0);
    if (int rc = gall(h, [&](kfsp_ctx *c, int p) {
            return kfsp_arnoldi(c, m, jold, qiop, break_tol, Hp[(size_t)p].data(), ldh, &mb[(size_t)p], &kk[(size_t)p], &av[(size_t)p]);
        }))
        return rc;
    for (int p = 1; p < g->n; ++p)
        if (mb[(size_t)p] != mb[0] || kk[(size_t)p] != kk[0] || !same_bits(&av[0], &av[(size_t)p], 1) ||
            !same_bits(Hp[0].data(), Hp[(size_t)p].data(), hsz))
            return gfail(h, 4002, "ranks disagree on the Hessenberg matrix");
    std::memcpy(H, Hp[0].data(), hsz * sizeof(double));
    *mbrkdwn = mb[0];
    *k1 = kk[0];
    *avnorm = av[0];
    return 0;
}

int group_combine(kfsp_ctx *h, int mx, double beta, const double *y, double *wsum)
{
    Group *g = h->group;
    std::vector<double> ws((size_t)g->n, 0.0);
    if (int rc = gall(h, [&](kfsp_ctx *c, int p) { return kfsp_combine(c, mx, beta, y, &ws[(size_t)p]); })) return rc;
    for (int p = 1; p < g->n; ++p)
        if (!same_bits(&ws[0], &ws[(size_t)p], 1)) return gfail(h, 4002, "ranks disagree on WSUM");
    *wsum = ws[0];
    return 0;
}

int group_restore_w(kfsp_ctx *h, double beta)
{
    return gall(h, [&](kfsp_ctx *c, int) { return kfsp_restore_w(c, beta); });
}

int group_spmv(kfsp_ctx *h, const double *x, double *y)
{
    return gall(h, [&](kfsp_ctx *c, int p) { return kfsp_spmv(c, x, y ? y + block_of(h, h->n, p).row0 : y); });
}

int group_spmv_w(kfsp_ctx *h, double *y)
{
    return gall(h, [&](kfsp_ctx *c, int p) { return kfsp_spmv_w(c, y ? y + block_of(h, h->n, p).row0 : y); });
}

// (kfsp_onestep / kfsp_onestep_columns / kfsp_propensities: work on the whole lists, no collective inside - the entry
// points hand a head's call to rank 0, group_rank0)

int group_drop_plan(kfsp_ctx *h, double dsum, double *droptol, int64_t *drop_count, int64_t *n_flagged)
{
    Group *g = h->group;
    std::vector<double> tol((size_t)g->n, 0.0);
    std::vector<int64_t> cnt((size_t)g->n, 0), nf((size_t)g->n, 0);
    if (int rc = gall(h, [&](kfsp_ctx *c, int p) { return kfsp_drop_plan(c, dsum, &tol[(size_t)p], &cnt[(size_t)p], &nf[(size_t)p]); }))
        return rc;
    for (int p = 1; p < g->n; ++p)
        if (!same_bits(&tol[0], &tol[(size_t)p], 1) || cnt[(size_t)p] != cnt[0] || nf[(size_t)p] != nf[0])
            return gfail(h, 4002, "ranks disagree on the drop plan");
    *droptol = tol[0];
    *drop_count = cnt[0];
    *n_flagged = nf[0];
    return 0;
}

int group_drop_flags(kfsp_ctx *h, int64_t n, uint8_t *dropped)
{
    // every rank holds the flags of all states; one copy is enough
    kfsp_ctx *c = h->group->sub[0];
    const int rc = kfsp_drop_flags(c, n, dropped);
    if (rc) h->err = kfsp_last_error(c);
    return rc;
}

int group_drop_compact(kfsp_ctx *h, int64_t *n_new)
{
    Group *g = h->group;
    std::vector<int64_t> nn((size_t)g->n, 0);
    if (int rc = gall(h, [&](kfsp_ctx *c, int p) { return kfsp_drop_compact(c, &nn[(size_t)p]); })) return rc;
    for (int p = 1; p < g->n; ++p)
        if (nn[(size_t)p] != nn[0]) return gfail(h, 4002, "ranks disagree on the compacted size");
    *n_new = nn[0];
    return 0;
}

int group_drop_rebuild(kfsp_ctx *h)
{
    int rc = gall(h, [&](kfsp_ctx *c, int) { return kfsp_drop_rebuild(c); });
    int64_t n = 0;
    if (!rc) rc = kfsp_num_states(h->group->sub[0], &n);
    return after_matrix(h, rc, n);
}

int group_expand_resident(kfsp_ctx *h, double t_ssa, int64_t seedmix, int32_t ns, int32_t nr, const int32_t *stoich, int32_t max_count,
                          int32_t capacity, int64_t *n_new, int64_t *n_from_ssa)
{
    Group *g = h->group;
    std::vector<int64_t> nn((size_t)g->n, 0), ns_((size_t)g->n, 0);
    int rc = gall(h, [&](kfsp_ctx *c, int p) {
        return kfsp_expand_resident(c, t_ssa, seedmix, ns, nr, stoich, max_count, capacity, &nn[(size_t)p], &ns_[(size_t)p]);
    });
    if (!rc)
        for (int p = 1; p < g->n; ++p)
            if (nn[(size_t)p] != nn[0] || ns_[(size_t)p] != ns_[0]) return gfail(h, 4002, "ranks disagree on the expanded FSP");
    if (!rc) {
        *n_new = nn[0];
        if (n_from_ssa) *n_from_ssa = ns_[0];
    }
    return after_matrix(h, rc, nn[0]);
}

int group_reduce_w(kfsp_ctx *h, int squared, double *out)
{
    Group *g = h->group;
    std::vector<double> v((size_t)g->n, 0.0);
    if (int rc = gall(h, [&](kfsp_ctx *c, int p) { return squared ? kfsp_nrm2_w(c, &v[(size_t)p]) : kfsp_asum_w(c, &v[(size_t)p]); }))
        return rc;
    for (int p = 1; p < g->n; ++p)
        if (!same_bits(&v[0], &v[(size_t)p], 1)) return gfail(h, 4002, "ranks disagree on a norm");
    *out = v[0];
    return 0;
}

int group_get_basis(kfsp_ctx *h, int j, int64_t n, double *v)
{
    if (n != h->n) return gfail(h, -3, "n is not the size of the FSP");
    return gall(h, [&](kfsp_ctx *c, int p) {
        const Block b = block_of(h, n, p);
        return kfsp_get_basis(c, j, b.nrows, v ? v + b.row0 : v);
    });
}

int group_spmv_bench(kfsp_ctx *h, int reps, int variant, float *ms_total)
{
    Group *g = h->group;
    std::vector<float> ms((size_t)g->n, 0.f);
    if (int rc = gall(h, [&](kfsp_ctx *c, int p) { return kfsp_spmv_bench(c, reps, variant, &ms[(size_t)p]); })) return rc;
    float t = 0.f;
    for (float x : ms) t = std::max(t, x);
    *ms_total = t;
    return 0;
}

int group_set_propensity_program(kfsp_ctx *h, int32_t ns, int32_t nr, int32_t np, const double *params, const int32_t *code_off,
                                 const int32_t *code, const int32_t *imm_off, const double *imm, const int32_t *tab_species,
                                 int32_t tab_len, const double *tab)
{
    return gall(h, [&](kfsp_ctx *c, int) {
        return kfsp_set_propensity_program(c, ns, nr, np, params, code_off, code, imm_off, imm, tab_species, tab_len, tab);
    });
}

int group_set_propensity_tables2(kfsp_ctx *h, int32_t nr, const int32_t *s1, const int32_t *s2, const int32_t *n1, const int32_t *n2,
                                 const int64_t *off, int64_t len, const double *tab2)
{
    return gall(h, [&](kfsp_ctx *c, int) { return kfsp_set_propensity_tables2(c, nr, s1, s2, n1, n2, off, len, tab2); });
}

kfsp_ctx *group_rank0(const kfsp_ctx *h) { return h->group->sub[0]; }

int group_layout_info(const kfsp_ctx *h, int64_t *v)
{
    // format, exchange and halo of rank 0 (agreed by all ranks); reach = max, chunks / coded chunks / code words = sums
    int64_t t[8];
    if (int rc = kfsp_layout_info(h->group->sub[0], v)) return rc;
    for (int p = 1; p < h->group->n; ++p) {
        if (int rc = kfsp_layout_info(h->group->sub[(size_t)p], t)) return rc;
        v[3] = std::max(v[3], t[3]);
        v[4] += t[4];
        v[5] += t[5];
        v[6] += t[6];
    }
    return 0;
}

int group_get_timers(kfsp_ctx *h, double *ms, int reset)
{
    // device phases from rank 0 (all ranks run in lock step), host phases (Pade, callbacks) from the head itself
    double t0[KFSP_T_COUNT];
    if (int rc = kfsp_get_timers(h->group->sub[0], t0, reset)) return rc;
    if (reset)
        for (int p = 1; p < h->group->n; ++p) {
            double dump[KFSP_T_COUNT];
            (void)kfsp_get_timers(h->group->sub[(size_t)p], dump, 1);
        }
    for (int i = 0; i < KFSP_T_COUNT; ++i) ms[i] = t0[i] + h->t_ms[i];
    if (reset)
        for (double &t : h->t_ms) t = 0.0;
    return 0;
}

// The watchdog on its own, no device anywhere (tests/test_group_watchdog.py, runs without a GPU): `nranks` worker threads
// and a loop-back transport; every rank sleeps work_ms, rank `failing_rank` (or none: -1) then returns -77 BEFORE the
// collective, rank `hanging_rank` (or none) sleeps hang_ms instead of entering it, the others enter one loop-back barrier
// (returning 2999 when it is released by the abort).  Out: the code run() returned, the rank it blamed, the seconds it
// took, and whether the group ended broken / stuck.
int group_selftest(int nranks, int failing_rank, int hanging_rank, int work_ms, int hang_ms, int timeout_ms, int grace_ms,
                   int settle_ms, int *rc_out, int *who_out, double *seconds, int *broken, int *stuck)
{
    if (nranks < 1 || nranks > 64 || !rc_out || !who_out || !seconds || !broken || !stuck) return -1;
    Group *g = new (std::nothrow) Group;       // (leaked when a rank is left behind - it may still touch it)
    LoopGroup *loop = new (std::nothrow) LoopGroup;
    if (!g || !loop) return 4001;
    g->n = nranks;
    g->rc.assign((size_t)nranks, 0);
    g->done.assign((size_t)nranks, 0);
    g->timeout_s = 1e-3 * timeout_ms;
    g->grace_s = 1e-3 * grace_ms;
    g->settle_s = 1e-3 * settle_ms;
    loop->n = nranks;
    loop->slot.assign((size_t)nranks, nullptr);
    g->abort_all = [loop] { loop->abort(); };
    for (int p = 0; p < nranks; ++p) g->th.emplace_back(&Group::worker, g, p);
    const auto t0 = std::chrono::steady_clock::now();
    int who = -1;
    const int rc = g->run([&](int p) {
        std::this_thread::sleep_for(std::chrono::milliseconds(work_ms));
        if (p == failing_rank) return -77;
        if (p == hanging_rank) {
            std::this_thread::sleep_for(std::chrono::milliseconds(hang_ms));
            return 0;
        }
        return loop->barrier() ? 0 : 2999;
    }, &who);
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *rc_out = rc;
    *who_out = who;
    *broken = g->broken ? 1 : 0;
    *stuck = g->stuck ? 1 : 0;
    // a second call on a broken group must come back at once
    if (g->broken && g->run([](int) { return 0; }) == 0) return -2;
    if (g->stuck) {
        for (std::thread &t : g->th) t.detach();
        return 0;
    }
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
    }
    g->cv_go.notify_all();
    for (std::thread &t : g->th) t.join();
    delete loop;
    delete g;
    return 0;
}

}  // namespace kfsp
