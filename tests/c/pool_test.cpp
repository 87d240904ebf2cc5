// CPU test of libaec_amd/csrc/aec_pool.h (the host threads of the batch entry points): every task of every job runs
// exactly once, jobs from two caller threads at a time (the second finds the pool busy and runs on threads of its
// own), task counts from 1 to 12 (more than the pool keeps workers), and a forked child gets a pool of its own.
//   g++ -O1 -g -fsanitize=thread -std=c++17 -pthread pool_test.cpp -o pool_test && ./pool_test
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <sys/wait.h>
#include <unistd.h>

#include "../../libaec_amd/csrc/aec_pool.h"

static int one_caller(unsigned seed, int jobs)
{
    int bad = 0;
    for (int j = 0; j < jobs; j++) {
        seed = seed * 1664525u + 1013904223u;
        const size_t count = 1 + (seed >> 24) % 12;
        std::vector<std::atomic<int>> hits(count);
        for (auto &h : hits) h = 0;
        std::atomic<long> sum{0};
        aec::WorkerPool::run(count, [&](size_t i) {
            hits[i]++;
            long s = 0;
            for (int k = 0; k < 2000 + (int)(i * 500); k++) s += k % 7;
            sum += s;
        });
        for (size_t i = 0; i < count; i++) bad += hits[i] != 1;
    }
    return bad;
}

// a job started from inside a task: runs in turn on the task's thread, every task once
static int nested(int jobs)
{
    int bad = 0;
    for (int j = 0; j < jobs; j++) {
        const size_t outer = 2 + j % 5, inner = 1 + j % 7;
        std::vector<std::atomic<int>> hits(outer * inner);
        for (auto &h : hits) h = 0;
        aec::WorkerPool::run(outer, [&](size_t i) {
            aec::WorkerPool::run(inner, [&](size_t k) { hits[i * inner + k]++; });
        });
        for (auto &h : hits) bad += h != 1;
    }
    return bad;
}

int main()
{
    int bad = one_caller(1, 300);
    bad += nested(200);
    std::atomic<int> bad2{0};
    std::thread a([&] { bad2 += one_caller(2, 400); }), b([&] { bad2 += one_caller(3, 400); });
    a.join();
    b.join();
    bad += bad2;
    const pid_t pid = fork();
    if (pid == 0) _exit(one_caller(4, 50) ? 1 : 0);
    int st = 0;
    waitpid(pid, &st, 0);
    bad += !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
    bad += one_caller(5, 100);
    printf("pool test: %d problems\n", bad);
    return bad ? 1 : 0;
}
