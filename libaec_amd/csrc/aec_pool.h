// aec_pool.h -- the host threads of the batch entry points (aec_abi.cpp), kept between calls.
//
// A batch of chunks runs as up to four parts side by side, every part driven by its own host thread (its own HIP
// stream and buffers).  The threads were created and joined per call; a per-chunk caller such as HDF5's filter
// pipeline makes thousands of such calls.  Here they are started once and sleep on a condition variable in
// between.  One job at a time: a caller that finds the pool busy (another user thread is inside a batch call)
// runs its job on threads of its own, as before.
//
// The pool is a heap object that is never destroyed (no destructor runs at exit while workers sleep in it), the
// library is linked with -z nodelete (a dlclose must not unmap code that sleeping threads return into), and a
// forked child, which has none of the threads, starts with a fresh pool.
#pragma once

#include <pthread.h>

#include <new>

#include <condition_variable>
#include <cstddef>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace aec {

class WorkerPool {
public:
    // f(0) ... f(count - 1), f(0) on the calling thread, the others on pool threads (or the calling thread, if
    // they are not taken in time); returns when all have returned.  f must not throw.
    static void run(size_t count, const std::function<void(size_t)> &f)
    {
        if (count == 0) return;
        if (count == 1) {
            f(0);
            return;
        }
        // a job started from inside a task (of the pool or of run_on_new_threads): in turn on this thread -- the
        // pool's lock may be held by this very thread (try_lock on a mutex its owner holds is undefined), and a
        // worker that waited for other workers could wait for itself
        if (inside()) {
            for (size_t i = 0; i < count; i++) f(i);
            return;
        }
        struct Mark {
            Mark() { inside() = true; }
            ~Mark() { inside() = false; }
        } mark;
        WorkerPool *p = instance();
        if (!p || !p->busy_.try_lock()) {
            run_on_new_threads(count, f);
            return;
        }
        p->job(count, f);
        p->busy_.unlock();
    }

private:
    static constexpr size_t kMaxWorkers = 8;
    std::mutex busy_;                        // one job at a time
    std::mutex mu_;
    std::condition_variable work_, done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0, next_ = 0, finished_ = 0, workers_ = 0;

    static bool &inside()                    // this thread is running a task of some job (or waiting for one's tasks)
    {
        static thread_local bool in = false;
        return in;
    }
    static WorkerPool *&slot()
    {
        static WorkerPool *p = nullptr;
        return p;
    }
    static std::mutex &slot_mu()
    {
        static std::mutex *m = new std::mutex;           // (never destroyed)
        return *m;
    }
    // fork: the child has none of the threads (the old object leaks there); the lock that guards the pointer is held
    // across the fork so that the child does not inherit it locked by a thread it does not have
    static void fork_prepare() { slot_mu().lock(); }
    static void fork_parent() { slot_mu().unlock(); }
    static void fork_child()
    {
        slot() = nullptr;
        slot_mu().unlock();
    }
    static WorkerPool *instance()
    {
        std::lock_guard<std::mutex> lk(slot_mu());
        if (!slot()) {
            static bool hooked = false;
            if (!hooked) {
                hooked = true;
                (void)pthread_atfork(fork_prepare, fork_parent, fork_child);
            }
            slot() = new (std::nothrow) WorkerPool;
        }
        return slot();
    }

    void worker()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            work_.wait(lk, [&] { return fn_ && next_ < n_; });
            const size_t i = next_++;
            const std::function<void(size_t)> *f = fn_;
            lk.unlock();
            inside() = true;
            (*f)(i);
            lk.lock();
            if (++finished_ == n_) done_.notify_all();
        }
    }

    void job(size_t count, const std::function<void(size_t)> &f)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &f;
            n_ = count;
            next_ = 1;                       // (task 0 is the caller's)
            finished_ = 0;
            while (workers_ < count - 1 && workers_ < kMaxWorkers) {
                try {
                    std::thread(&WorkerPool::worker, this).detach();
                } catch (...) {
                    break;                   // (no more threads: the caller takes what the workers do not)
                }
                workers_++;
            }
        }
        work_.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu_);
        finished_++;
        // whatever no worker has taken yet, the caller takes itself
        while (next_ < n_) {
            const size_t i = next_++;
            lk.unlock();
            f(i);
            lk.lock();
            finished_++;
        }
        done_.wait(lk, [&] { return finished_ == n_; });
        fn_ = nullptr;
        n_ = next_ = 0;
    }

    static void run_on_new_threads(size_t count, const std::function<void(size_t)> &f)
    {
        std::vector<std::thread> th;
        size_t started = 1;
        try {
            for (; started < count; started++)
                th.emplace_back([&f](size_t i) {
                    inside() = true;
                    f(i);
                }, started);
        } catch (...) {                      // (no more threads: this one does the rest in turn)
        }
        f(0);
        for (size_t t = started; t < count; t++) f(t);
        for (std::thread &x : th) x.join();
    }
};

}  // namespace aec
