/*
 * cpu_bench.c -- the timed CPU baseline of bench.py (SURVEY.md 8d "CPU baseline beside it"): the oracle's
 * single-threaded-per-core BVH traversal + shading (pt_oracle.c, kind "port"), driven by one PINNED native thread per
 * physical core.  TEST / MEASUREMENT INFRASTRUCTURE, not product code: only bench.py's cpu_baseline leg loads
 * libcpubench.so.  Built -O3 -march=native -fno-fast-math -ffp-contract=off (oracle/Makefile), so the arithmetic -- and
 * every pixel -- is that of the checker build; only the code generation differs.
 *
 * Work distribution: the frame is cut into 4-row bands handed out by one atomic counter, frame after frame, until the
 * time budget of a repetition is spent (checked at band granularity; the last frame is finished).  Three repetitions,
 * the median rate is reported ("median of 3").
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pt_oracle.c"

typedef struct OrcBenchResult {
    double   mray_per_s;       /* median over the repetitions */
    double   rep_mray_per_s[3];
    double   seconds;          /* wall time of all repetitions */
    double   frames;           /* frames rendered in all repetitions */
    uint64_t rays;             /* closest + shadow rays traced in all repetitions */
    int32_t  threads;
    int32_t  pinned;           /* threads whose affinity call succeeded */
} OrcBenchResult;

typedef struct BenchShared {
    const OrcScene* sc;
    const OrcParams* p;
    uint32_t max_depth, n_bands;
    atomic_ulong next;
    atomic_int   stop;
    unsigned long stop_at;       /* band index at which the repetition ends (set by the thread that sees the budget spent) */
    double t_start, budget;
    pthread_barrier_t bar;
} BenchShared;

typedef struct BenchThread {
    BenchShared* sh;
    int cpu, pinned;
    OrcStats stats;
    pthread_t th;
} BenchThread;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* bench_worker(void* arg)
{
    BenchThread* t = (BenchThread*)arg;
    BenchShared* sh = t->sh;
    if (t->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(t->cpu, &set);
        t->pinned = pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0;
    }
    pthread_barrier_wait(&sh->bar);
    for (;;) {
        const unsigned long b = atomic_fetch_add(&sh->next, 1ul);
        if (atomic_load(&sh->stop) && b >= sh->stop_at) break;
        if (b % sh->n_bands == 0 && b > 0 && !atomic_load(&sh->stop) && now_s() - sh->t_start > sh->budget) {
            /* the budget is spent: this frame boundary ends the repetition; bands below it are still rendered */
            sh->stop_at = b;
            atomic_store(&sh->stop, 1);
            break;
        }
        const uint32_t y = (uint32_t)(b % sh->n_bands) * 4u;
        const uint32_t y1 = y + 4u < sh->p->height ? y + 4u : sh->p->height;
        orc_render_rows(sh->sc, ORC_MODE_BVH, sh->p, sh->max_depth, y, y1, &t->stats);
    }
    return NULL;
}

static int cmp_double(const void* a, const void* b)
{
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* cpu_ids: n_threads logical CPU numbers to pin to (one per physical core, chosen by the caller), or NULL = unpinned */
int orc_bench_render(const OrcScene* sc, const OrcParams* p, uint32_t max_depth, int n_threads, const int* cpu_ids,
                     double budget_s, OrcBenchResult* out)
{
    if (!sc || !p || !out || n_threads < 1 || n_threads > 1024 || !p->accum_buffer) return 1;
    memset(out, 0, sizeof *out);
    out->threads = n_threads;
    BenchThread* th = (BenchThread*)calloc((size_t)n_threads, sizeof *th);
    if (!th) return 2;
    const double t_all = now_s();
    for (int rep = 0; rep < 3; ++rep) {
        BenchShared sh;
        memset(&sh, 0, sizeof sh);
        sh.sc = sc; sh.p = p; sh.max_depth = max_depth; sh.n_bands = (p->height + 3u) / 4u;
        atomic_init(&sh.next, 0ul); atomic_init(&sh.stop, 0);
        sh.stop_at = ~0ul; sh.budget = budget_s / 3.0;
        pthread_barrier_init(&sh.bar, NULL, (unsigned)n_threads + 1u);
        for (int i = 0; i < n_threads; ++i) {
            memset(&th[i].stats, 0, sizeof th[i].stats);
            th[i].sh = &sh; th[i].cpu = cpu_ids ? cpu_ids[i] : -1; th[i].pinned = 0;
            if (pthread_create(&th[i].th, NULL, bench_worker, &th[i]) != 0) { free(th); return 3; }
        }
        sh.t_start = now_s();
        pthread_barrier_wait(&sh.bar);
        sh.t_start = now_s();
        uint64_t rays = 0;
        int pinned = 0;
        for (int i = 0; i < n_threads; ++i) {
            pthread_join(th[i].th, NULL);
            rays += th[i].stats.closest_rays + th[i].stats.shadow_rays;
            pinned += th[i].pinned;
        }
        const double dt = now_s() - sh.t_start;
        pthread_barrier_destroy(&sh.bar);
        out->rep_mray_per_s[rep] = (double)rays / dt / 1e6;
        out->rays += rays;
        out->frames += (double)(sh.stop_at == ~0ul ? atomic_load(&sh.next) : sh.stop_at) / (double)sh.n_bands;
        out->pinned = pinned;
    }
    out->seconds = now_s() - t_all;
    double r[3] = { out->rep_mray_per_s[0], out->rep_mray_per_s[1], out->rep_mray_per_s[2] };
    qsort(r, 3, sizeof r[0], cmp_double);
    out->mray_per_s = r[1];
    free(th);
    return 0;
}
