/*
 * pool_selftest.c -- the legacy pool shim (csrc/host/threadpool.c) without a
 * GPU, under ThreadSanitizer and AddressSanitizer.  The one call into the GPU
 * path (ptgpu_pool_render_rows) and render_task are replaced by recording
 * stand-ins, so the test sees exactly which rows the shim hands over and when.
 * Built and run by tests/test_sanitizers.py:
 *   gcc -fsanitize=thread -Iinclude -Ipath_tracer_c_amd/csrc/host
 *       tests/c/pool_selftest.c csrc/host/{threadpool,vector}.c -lpthread
 * What is checked is the reference pool's behaviour for ordinary tasks
 * (src/threadpool.c:14-95: run at once, by up to num_threads workers, newest
 * first) plus the documented intent of wait / destroy.
 */
#include "path_tracer_api.h"
#include "pool_rows.h"

#include <stdatomic.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

static int failures = 0;
#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); \
            failures++;                                                        \
        }                                                                      \
    } while (0)

/* ---- stand-ins for the GPU side ------------------------------------------------ */
static pthread_mutex_t seen_lock = PTHREAD_MUTEX_INITIALIZER;
static render_args seen_rows[4096];
static size_t n_seen_rows = 0, n_flushes = 0;

void render_task(render_args *arg) { /* never run by the pool: recognised by address */
    (void)arg;
    CHECK(!"render_task must not be run on a pool thread");
}

void ptgpu_pool_render_rows(const render_args *rows, size_t n) {
    pthread_mutex_lock(&seen_lock);
    for (size_t i = 0; i < n && n_seen_rows < 4096; i++)
        seen_rows[n_seen_rows++] = rows[i];
    n_flushes++;
    pthread_mutex_unlock(&seen_lock);
}

/* ---- helpers ---------------------------------------------------------------------- */
static void nap_ms(long ms) {
    struct timespec t = {ms / 1000, (ms % 1000) * 1000000L};
    nanosleep(&t, NULL);
}

/* waits up to ~10 s for *flag to reach `want` */
static bool reaches(atomic_int *flag, int want) {
    for (int i = 0; i < 10000; i++) {
        if (atomic_load(flag) >= want)
            return true;
        nap_ms(1);
    }
    return false;
}

typedef struct {
    atomic_int *counter;
    atomic_int *gate;   /* when set: the task waits until *gate != 0 */
    atomic_int *inside; /* when set: incremented on entry */
    int tag;
    int *order;         /* when set: order[(*order_n)++] = tag, under seen_lock */
    size_t *order_n;
    long nap;
} job;

static void run_job(void *arg) {
    job *j = (job *)arg;
    if (j->inside)
        atomic_fetch_add(j->inside, 1);
    if (j->gate)
        while (atomic_load(j->gate) == 0)
            nap_ms(1);
    if (j->nap)
        nap_ms(j->nap);
    if (j->order) {
        pthread_mutex_lock(&seen_lock);
        j->order[(*j->order_n)++] = j->tag;
        pthread_mutex_unlock(&seen_lock);
    }
    if (j->counter)
        atomic_fetch_add(j->counter, 1);
}

/* ---- the checks ------------------------------------------------------------------ */

/* ordinary tasks start without threadpool_wait_for_tasks, as the reference's do */
static void tasks_run_before_wait(void) {
    threadpool pool;
    threadpool_init(&pool, 3);
    CHECK(pool.threads.size == 3 && pool.threads_running);
    atomic_int done = 0;
    job j = {.counter = &done};
    threadpool_add_task(&pool, run_job, &j);
    CHECK(reaches(&done, 1));
    threadpool_wait_for_tasks(&pool); /* nothing left: returns at once */
    threadpool_destroy(&pool);
}

/* num_threads tasks can be inside their functions at the same time */
static void tasks_run_concurrently(void) {
    enum { N = 4 };
    threadpool pool;
    threadpool_init(&pool, N);
    atomic_int inside = 0, gate = 0, done = 0;
    job j[N];
    for (int i = 0; i < N; i++) {
        j[i] = (job){.counter = &done, .gate = &gate, .inside = &inside};
        threadpool_add_task(&pool, run_job, &j[i]);
    }
    CHECK(reaches(&inside, N)); /* all four entered while none could leave */
    CHECK(atomic_load(&done) == 0);
    atomic_store(&gate, 1);
    threadpool_wait_for_tasks(&pool);
    CHECK(atomic_load(&done) == N);
    threadpool_destroy(&pool);
}

/* one worker, held inside a first task while k more are queued: k-1 ... 0
 * (src/threadpool.c:28 pops the newest; the same procedure is run on the
 * reference's own pool by tests/test_ref_pins.py) */
static void one_worker_pops_the_newest(void) {
    enum { K = 12 };
    threadpool pool;
    threadpool_init(&pool, 1);
    atomic_int inside = 0, gate = 0, done = 0;
    int order[K + 1];
    size_t order_n = 0;
    job first = {.counter = &done, .gate = &gate, .inside = &inside};
    threadpool_add_task(&pool, run_job, &first);
    CHECK(reaches(&inside, 1));
    job j[K];
    for (int i = 0; i < K; i++) {
        j[i] = (job){.counter = &done, .tag = i, .order = order, .order_n = &order_n};
        threadpool_add_task(&pool, run_job, &j[i]);
    }
    atomic_store(&gate, 1);
    threadpool_wait_for_tasks(&pool);
    CHECK(atomic_load(&done) == K + 1 && order_n == K);
    for (int i = 0; i < K; i++)
        CHECK(order[i] == K - 1 - i);
    threadpool_destroy(&pool);
}

/* wait returns only when every task has finished (the reference can return early) */
static void wait_means_all_done(void) {
    enum { N = 40 };
    threadpool pool;
    threadpool_init(&pool, 5);
    atomic_int done = 0;
    job j[N];
    for (int i = 0; i < N; i++) {
        j[i] = (job){.counter = &done, .nap = 1 + i % 4};
        threadpool_add_task(&pool, run_job, &j[i]);
    }
    threadpool_wait_for_tasks(&pool);
    CHECK(atomic_load(&done) == N);
    threadpool_wait_for_tasks(&pool);
    threadpool_destroy(&pool);
}

/* rows: snapshotted at add, none run on a thread, all handed over by one wait */
static void rows_go_to_the_gpu_path_at_wait(void) {
    enum { H = 50 };
    threadpool pool;
    threadpool_init(&pool, 2);
    static uint8_t frame[3 * 8 * H];
    scene world;
    memset(&world, 0, sizeof world);
    render_args a;
    n_seen_rows = n_flushes = 0;
    for (int y = 0; y < H; y++) {
        a = (render_args){.screen_y = y, .framew = 8, .frameh = H, .frame = frame, .world = &world};
        threadpool_add_task(&pool, (void (*)(void *))render_task, &a);
        a.screen_y = -77; /* the snapshot was taken: this must not show */
    }
    CHECK(pool.tasks.size == H && n_seen_rows == 0);
    threadpool_wait_for_tasks(&pool);
    CHECK(pool.tasks.size == 0 && n_seen_rows == H && n_flushes == 1);
    bool got[H] = {false};
    for (size_t i = 0; i < n_seen_rows; i++) {
        const render_args *r = &seen_rows[i];
        CHECK(r->screen_y >= 0 && r->screen_y < H && r->frame == frame && r->world == &world &&
              r->framew == 8 && r->frameh == H);
        if (r->screen_y >= 0 && r->screen_y < H) {
            CHECK(!got[r->screen_y]);
            got[r->screen_y] = true;
        }
    }
    /* a NULL argument cannot be a row: it is an ordinary task for render_task's address ...
       which this test does not queue (the stand-in render_task would flag it) */
    threadpool_destroy(&pool);
}

/* an ordinary task queues rows and more tasks: the same wait finishes them all */
typedef struct {
    threadpool *pool;
    render_args row;
    job inner;
} spawner;

static void spawn(void *arg) {
    spawner *s = (spawner *)arg;
    nap_ms(5);
    threadpool_add_task(s->pool, (void (*)(void *))render_task, &s->row);
    threadpool_add_task(s->pool, run_job, &s->inner);
}

static void tasks_may_queue_tasks(void) {
    threadpool pool;
    threadpool_init(&pool, 3);
    static uint8_t frame[3 * 4 * 4];
    scene world;
    memset(&world, 0, sizeof world);
    atomic_int done = 0;
    spawner s[6];
    n_seen_rows = n_flushes = 0;
    for (int i = 0; i < 6; i++) {
        s[i].pool = &pool;
        s[i].row = (render_args){.screen_y = i % 4, .framew = 4, .frameh = 4, .frame = frame, .world = &world};
        s[i].inner = (job){.counter = &done, .nap = 2};
        threadpool_add_task(&pool, spawn, &s[i]);
    }
    threadpool_wait_for_tasks(&pool);
    CHECK(atomic_load(&done) == 6 && n_seen_rows == 6 && pool.tasks.size == 0);
    threadpool_destroy(&pool);
}

/* destroy: running tasks finish, queued ones are dropped (src/threadpool.c:60-74),
 * and it returns with several idle workers (the reference signals only one) */
static void *destroy_it(void *arg) {
    threadpool_destroy((threadpool *)arg);
    return NULL;
}

static void destroy_drops_what_is_queued(void) {
    threadpool pool;
    threadpool_init(&pool, 2);
    atomic_int inside = 0, gate = 0, done = 0, dropped = 0;
    job held[2], later[5];
    for (int i = 0; i < 2; i++) {
        held[i] = (job){.counter = &done, .gate = &gate, .inside = &inside};
        threadpool_add_task(&pool, run_job, &held[i]);
    }
    CHECK(reaches(&inside, 2));
    for (int i = 0; i < 5; i++) {
        later[i] = (job){.counter = &dropped};
        threadpool_add_task(&pool, run_job, &later[i]);
    }
    static uint8_t frame[3 * 4 * 4];
    scene world;
    memset(&world, 0, sizeof world);
    render_args row = {.screen_y = 1, .framew = 4, .frameh = 4, .frame = frame, .world = &world};
    n_seen_rows = 0;
    threadpool_add_task(&pool, (void (*)(void *))render_task, &row);
    pthread_t t;
    pthread_create(&t, NULL, destroy_it, &pool);
    for (int i = 0; i < 10000; i++) { /* until destroy has said stop */
        pthread_mutex_lock(&pool.tasks_mutex);
        const bool stopping = !pool.threads_running;
        pthread_mutex_unlock(&pool.tasks_mutex);
        if (stopping)
            break;
        nap_ms(1);
    }
    atomic_store(&gate, 1);
    pthread_join(t, NULL);
    CHECK(atomic_load(&done) == 2 && atomic_load(&dropped) == 0 && n_seen_rows == 0);

    /* several idle workers, nothing queued */
    threadpool_init(&pool, 6);
    job quick[6];
    atomic_int q = 0;
    for (int i = 0; i < 6; i++) {
        quick[i] = (job){.counter = &q, .nap = 2};
        threadpool_add_task(&pool, run_job, &quick[i]);
    }
    threadpool_wait_for_tasks(&pool);
    CHECK(atomic_load(&q) == 6);
    threadpool_destroy(&pool);

    /* never used */
    threadpool_init(&pool, 4);
    threadpool_destroy(&pool);
}

/* num_threads == 0: one worker serves (the reference would never run the task) */
static void zero_threads_still_runs(void) {
    threadpool pool;
    threadpool_init(&pool, 0);
    atomic_int done = 0;
    job j = {.counter = &done};
    threadpool_add_task(&pool, run_job, &j);
    threadpool_add_task(&pool, NULL, NULL); /* ignored */
    threadpool_wait_for_tasks(&pool);
    CHECK(atomic_load(&done) == 1);
    threadpool_destroy(&pool);
}

/* many small tasks from several producers */
typedef struct {
    threadpool *pool;
    job *jobs;
    int n;
} producer;

static void *produce(void *arg) {
    producer *p = (producer *)arg;
    for (int i = 0; i < p->n; i++)
        threadpool_add_task(p->pool, run_job, &p->jobs[i]);
    return NULL;
}

static void many_tasks_many_producers(void) {
    enum { P = 4, N = 2500 };
    threadpool pool;
    threadpool_init(&pool, 8);
    static job jobs[P][N];
    atomic_int done = 0;
    producer prod[P];
    pthread_t t[P];
    for (int p = 0; p < P; p++) {
        for (int i = 0; i < N; i++)
            jobs[p][i] = (job){.counter = &done};
        prod[p] = (producer){&pool, jobs[p], N};
        pthread_create(&t[p], NULL, produce, &prod[p]);
    }
    for (int p = 0; p < P; p++)
        pthread_join(t[p], NULL);
    threadpool_wait_for_tasks(&pool);
    CHECK(atomic_load(&done) == P * N);
    threadpool_destroy(&pool);
}

int main(void) {
    tasks_run_before_wait();
    tasks_run_concurrently();
    one_worker_pops_the_newest();
    wait_means_all_done();
    rows_go_to_the_gpu_path_at_wait();
    tasks_may_queue_tasks();
    destroy_drops_what_is_queued();
    zero_threads_still_runs();
    many_tasks_many_producers();
    if (failures) {
        fprintf(stderr, "pool_selftest: %d failure(s)\n", failures);
        return 1;
    }
    printf("pool_selftest: ok\n");
    return 0;
}
