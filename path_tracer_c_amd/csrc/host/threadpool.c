// threadpool_* of the legacy surface (include/path_tracer_api.h), replacing
// src/threadpool.c:14-101 of the reference.
//
// The reference's pool hands every task to one of `num_threads` pthreads, which
// pop the NEWEST task (src/threadpool.c:28, src/vector.c:24-31).  Here a task
// whose function is render_task (pointer identity) is not run on a thread at
// all: its row descriptor is snapshotted and queued, and threadpool_wait_for_tasks
// hands all queued rows to the GPU path in one go (pool_rows.h).  Any other
// ("foreign") task is run the reference's way: by up to `num_threads` worker
// pthreads, started when the first foreign task arrives, newest task first, as
// soon as a worker is free -- a caller that queues its own work beside the rows,
// or that waits for a task's effect before calling threadpool_wait_for_tasks,
// sees what it saw with the reference.
//
// Differences, all on the side of the documented intent:
//  * threadpool_wait_for_tasks returns when EVERY task queued so far has
//    finished (the reference waits on a condition variable without a predicate:
//    it can return while tasks are still running, or miss its wake-up);
//  * threadpool_destroy wakes ALL workers (the reference signals one and can
//    hang in pthread_join with more than one idle worker);
//  * num_threads == 0 is served by one worker (the reference would never run
//    the tasks);
//  * allocation failures are checked.
//
// State that does not fit the caller-allocated struct hangs off it:
// pool->threads holds `num_threads` pool_worker records (first member: the
// pthread_t, so *(pthread_t *)threads.data[i] reads as in the reference once the
// worker has been started); record 0 is a pool_head, which also carries the
// foreign-task stack and the count of foreign tasks being executed.  Everything
// is guarded by pool->tasks_mutex.
#include "path_tracer_api.h"
#include "pool_rows.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    void (*fn)(void *);
    void *arg;
    bool is_row;
    render_args row; // snapshot taken when the task was added (src/renderer.c:93 copies it at entry)
} pool_task;

typedef struct {
    pthread_t thread;
    bool started;
} pool_worker;

typedef struct {
    pool_worker worker; // must stay first: threads.data[0] is read as a pool_worker
    vector foreign;     // pool_task*, newest last
    size_t running;     // foreign tasks inside their function right now
    size_t started;     // workers created so far
} pool_head;

static void *checked_alloc(size_t bytes) {
    void *p = calloc(1, bytes);
    if (p == NULL) {
        fprintf(stderr, "libptgpu: threadpool: out of memory (%zu bytes)\n", bytes);
        abort();
    }
    return p;
}

static pool_head *head_of(threadpool *pool) {
    return (pool->threads.size > 0) ? (pool_head *)pool->threads.data[0] : NULL;
}

static void *worker_main(void *arg) {
    threadpool *pool = (threadpool *)arg;
    pthread_mutex_lock(&pool->tasks_mutex);
    pool_head *head = head_of(pool);
    for (;;) {
        if (head == NULL) // (cannot be: workers are started by an initialised pool)
            break;
        while (pool->threads_running && head->foreign.size == 0)
            pthread_cond_wait(&pool->tasks_available_cond, &pool->tasks_mutex);
        if (!pool->threads_running)
            break; // destroy: what is still queued is dropped (src/threadpool.c:70-74)
        pool_task *task = (pool_task *)vector_pop(&head->foreign);
        head->running++;
        pthread_mutex_unlock(&pool->tasks_mutex);
        task->fn(task->arg);
        free(task);
        pthread_mutex_lock(&pool->tasks_mutex);
        head->running--;
        if (head->running == 0 && head->foreign.size == 0)
            pthread_cond_broadcast(&pool->tasks_exhausted_cond);
    }
    pthread_mutex_unlock(&pool->tasks_mutex);
    return NULL;
}

// called with tasks_mutex held: one more worker while there are more waiting
// foreign tasks than idle workers
static void start_workers_as_needed(threadpool *pool, pool_head *head) {
    while (head->started < pool->threads.size &&
           head->foreign.size > head->started - head->running) {
        pool_worker *w = (pool_worker *)pool->threads.data[head->started];
        if (pthread_create(&w->thread, NULL, worker_main, pool) != 0)
            break; // no thread to be had: threadpool_wait_for_tasks runs the task itself
        w->started = true;
        head->started++;
    }
}

void threadpool_init(threadpool *pool, size_t num_threads) {
    const size_t n = num_threads ? num_threads : 1;
    vector_init_with_capacity(&pool->threads, n);
    vector_init(&pool->tasks);
    pthread_mutex_init(&pool->tasks_mutex, NULL);
    pthread_cond_init(&pool->tasks_available_cond, NULL);
    pthread_cond_init(&pool->tasks_exhausted_cond, NULL);
    pool->threads_running = true;
    pool_head *head = (pool_head *)checked_alloc(sizeof *head);
    vector_init(&head->foreign);
    vector_push(&pool->threads, head);
    for (size_t i = 1; i < n; i++)
        vector_push(&pool->threads, checked_alloc(sizeof(pool_worker)));
}

void threadpool_add_task(threadpool *pool, void (*f)(void *), void *arg) {
    if (f == NULL)
        return;
    pool_task *task = (pool_task *)checked_alloc(sizeof *task);
    task->fn = f;
    task->arg = arg;
    task->is_row = (f == (void (*)(void *))render_task) && arg != NULL;
    if (task->is_row)
        task->row = *(const render_args *)arg;
    pthread_mutex_lock(&pool->tasks_mutex);
    if (task->is_row) {
        vector_push(&pool->tasks, task);
    } else {
        pool_head *head = head_of(pool);
        if (head == NULL) { // a pool that was never initialised (or already destroyed)
            free(task);
        } else {
            vector_push(&head->foreign, task);
            start_workers_as_needed(pool, head);
            pthread_cond_signal(&pool->tasks_available_cond);
        }
    }
    pthread_mutex_unlock(&pool->tasks_mutex);
}

void threadpool_wait_for_tasks(threadpool *pool) {
    for (;;) {
        // every row queued so far, to the GPU in one go
        pthread_mutex_lock(&pool->tasks_mutex);
        const size_t n = pool->tasks.size;
        render_args *rows = n ? (render_args *)checked_alloc(n * sizeof *rows) : NULL;
        for (size_t i = 0; i < n; i++) {
            pool_task *t = (pool_task *)vector_pop(&pool->tasks); // newest first
            rows[i] = t->row;
            free(t);
        }
        pthread_mutex_unlock(&pool->tasks_mutex);
        if (n) {
            ptgpu_pool_render_rows(rows, n);
            free(rows);
        }

        // the foreign tasks: the workers are on them; wait until none is queued
        // or running.  (Without a single worker -- pthread_create failed -- they
        // run here, newest first.)
        pthread_mutex_lock(&pool->tasks_mutex);
        pool_head *head = head_of(pool);
        if (head == NULL) { // never initialised: nothing can be queued
            pthread_mutex_unlock(&pool->tasks_mutex);
            return;
        }
        while (head->started == 0 && head->foreign.size != 0) {
            pool_task *t = (pool_task *)vector_pop(&head->foreign);
            pthread_mutex_unlock(&pool->tasks_mutex);
            t->fn(t->arg);
            free(t);
            pthread_mutex_lock(&pool->tasks_mutex);
        }
        while (head->running != 0 || head->foreign.size != 0)
            pthread_cond_wait(&pool->tasks_exhausted_cond, &pool->tasks_mutex);
        // a foreign task may have queued rows meanwhile
        const bool done = pool->tasks.size == 0;
        pthread_mutex_unlock(&pool->tasks_mutex);
        if (done)
            return;
    }
}

void threadpool_destroy(threadpool *pool) {
    pthread_mutex_lock(&pool->tasks_mutex);
    pool->threads_running = false;
    pthread_cond_broadcast(&pool->tasks_available_cond);
    pool_head *head = head_of(pool);
    const size_t started = head ? head->started : 0;
    pthread_mutex_unlock(&pool->tasks_mutex);
    // workers finish the task they are in and leave; nothing new is started
    for (size_t i = 0; i < started; i++)
        pthread_join(((pool_worker *)pool->threads.data[i])->thread, NULL);
    // unexecuted tasks are dropped, as in src/threadpool.c:70-74
    for (void *t; (t = vector_pop(&pool->tasks)) != NULL;)
        free(t);
    if (head) {
        for (void *t; (t = vector_pop(&head->foreign)) != NULL;)
            free(t);
        vector_free(&head->foreign);
    }
    for (size_t i = 0; i < pool->threads.size; i++)
        free(pool->threads.data[i]);
    vector_free(&pool->threads);
    vector_free(&pool->tasks);
    pthread_mutex_destroy(&pool->tasks_mutex);
    pthread_cond_destroy(&pool->tasks_available_cond);
    pthread_cond_destroy(&pool->tasks_exhausted_cond);
}
