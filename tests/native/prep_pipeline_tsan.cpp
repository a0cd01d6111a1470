// CPU race test of the host entry points' producer / consumer skeleton (manga-ocr_amd/csrc/prep_pipeline.h) with a stub
// device layer: "events" are atomics a fake copy engine sets some time after they were recorded, "lanes" finish their
// jobs after a delay.  Built with g++ -fsanitize=thread by tests/test_host_pipeline_tsan.py (TSAN build of
// the host scheduler).  Checked invariants:
//   * a job is pushed only after its chunk's preparation was enqueued, in chunk order, exactly once;
//   * pinned buffer k & 1 is repacked only after the copy of chunk k - 2 has completed;
//   * a job "starts" on the device only after its chunk's event completed (the device-side wait);
//   * an exception in the producer or in the pump stops the other side, is re-thrown on the caller, and on_abort runs.
#include <atomic>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <vector>

#include "../../manga-ocr_amd/csrc/prep_pipeline.h"

struct Stub {
    int nchunks;
    std::vector<std::atomic<int>> prepared, event_done, pushed, started;
    std::vector<std::thread> engine;          // the fake copy engine: one timer per recorded event
    std::mutex mu;
    std::vector<int> queue;                   // jobs pushed, not yet running (calling thread only)
    int lanes_busy = 0;                       // calling thread only
    std::vector<std::chrono::steady_clock::time_point> lane_end;
    int aborted = 0;
    explicit Stub(int n) : nchunks(n), prepared(n), event_done(n), pushed(n), started(n) {
        for (int i = 0; i < n; ++i) { prepared[i] = 0; event_done[i] = 0; pushed[i] = 0; started[i] = 0; }
    }
    ~Stub() { for (auto& t : engine) t.join(); }
};

static int run_case(int nchunks, int fail_prepare_at, int fail_pump_at, unsigned seed) {
    Stub s(nchunks);
    std::mt19937 rng(seed);
    auto us = [&](int lo, int hi) { return std::chrono::microseconds(lo + (int)(rng() % (unsigned)(hi - lo + 1))); };
    int pump_calls = 0;
    bool threw = false;
    try {
        run_prep_pipeline(
            nchunks,
            [&](int k) {                                           // producer: slot k & 1 free?
                if (k >= 2) {
                    while (!s.event_done[k - 2].load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
            },
            [&](int k) {                                           // producer: pack + enqueue + record
                if (k >= 2 && !s.event_done[k - 2].load()) { fprintf(stderr, "slot %d repacked while chunk %d's copy is in flight\n", k & 1, k - 2); abort(); }
                if (k == fail_prepare_at) throw std::runtime_error("prepare failed");
                std::this_thread::sleep_for(std::chrono::microseconds(200 + 97 * (k % 3)));       // packing
                s.prepared[k].store(1, std::memory_order_release);
                std::lock_guard<std::mutex> lk(s.mu);
                s.engine.emplace_back([&s, k] {                    // the copy + resize complete some time later
                    std::this_thread::sleep_for(std::chrono::microseconds(300 + 131 * (k % 4)));
                    s.event_done[k].store(1, std::memory_order_release);
                });
            },
            [&](int k) {                                           // caller: job k queued behind its event
                if (!s.prepared[k].load(std::memory_order_acquire)) { fprintf(stderr, "job %d pushed before its preparation\n", k); abort(); }
                if (s.pushed[k].fetch_add(1) != 0) { fprintf(stderr, "job %d pushed twice\n", k); abort(); }
                if (k > 0 && !s.pushed[k - 1].load()) { fprintf(stderr, "job %d pushed out of order\n", k); abort(); }
                s.queue.push_back(k);
            },
            [&]() -> bool {                                        // caller: two lanes
                if (++pump_calls == fail_pump_at) throw std::runtime_error("pump failed");
                const auto now = std::chrono::steady_clock::now();
                for (size_t i = 0; i < s.lane_end.size();)
                    if (s.lane_end[i] <= now) { s.lane_end.erase(s.lane_end.begin() + i); } else ++i;
                while (s.lane_end.size() < 2 && !s.queue.empty()) {
                    const int k = s.queue.front();
                    if (!s.event_done[k].load(std::memory_order_acquire)) break;      // the device-side wait for the chunk's event
                    s.queue.erase(s.queue.begin());
                    s.started[k].store(1);
                    s.lane_end.push_back(now + us(400, 900));
                }
                if (!s.lane_end.empty() || !s.queue.empty()) { std::this_thread::sleep_for(std::chrono::microseconds(60)); return true; }
                return false;
            },
            [&] { s.aborted += 1; });
    } catch (const std::runtime_error&) {
        threw = true;
    }
    const bool want_throw = (fail_prepare_at >= 0 && fail_prepare_at < nchunks) || fail_pump_at > 0;
    if (threw != want_throw) { fprintf(stderr, "case (%d, %d, %d): threw=%d\n", nchunks, fail_prepare_at, fail_pump_at, threw); return 1; }
    if (threw) { if (s.aborted != 1) { fprintf(stderr, "on_abort ran %d times\n", s.aborted); return 1; } return 0; }
    for (int k = 0; k < nchunks; ++k)
        if (s.pushed[k] != 1 || s.started[k] != 1) { fprintf(stderr, "chunk %d: pushed %d started %d\n", k, (int)s.pushed[k], (int)s.started[k]); return 1; }
    return 0;
}

int main() {
    int bad = 0;
    for (unsigned seed = 0; seed < 6; ++seed) {
        bad += run_case(2, -1, -1, seed);
        bad += run_case(7, -1, -1, seed);
        bad += run_case(16, -1, -1, seed);
    }
    bad += run_case(5, 0, -1, 1);       // the very first preparation fails
    bad += run_case(9, 4, -1, 2);       // a later one fails while jobs are in flight
    bad += run_case(9, -1, 7, 3);       // the pump (a HIP error on the calling thread) fails while the producer is packing
    bad += run_case(3, -1, 1, 4);
    printf(bad ? "FAILED %d\n" : "prep_pipeline: all cases passed\n", bad);
    return bad ? 1 : 0;
}
