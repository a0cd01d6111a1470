// Producer / consumer skeleton of the host entry points (mocr_recognize_images / _regions), free of any HIP call so
// that it can be built and raced on a CPU (tests/native/prep_pipeline_tsan.cpp: g++ -fsanitize=thread).
//
//   producer thread:  for k = 0 .. nchunks-1:  wait_slot(k)   the pinned buffer k & 1 is free again (chunk k-2's copy done)
//                                              prepare(k)     pack + enqueue H2D / resize of chunk k, record its event
//                                              -> chunk k is "enqueued"
//   calling thread:   pushes job k as soon as chunk k is enqueued (the job waits ON THE DEVICE for the chunk's event),
//                     keeps pumping the lanes, sleeps on the condition variable only while every lane is idle
//
// The calling thread holds the engine mutex for the whole call; the producer touches only what the callables capture
// (the preparation stream, the pinned buffers, the resample-table cache) - nothing the pump reads.
// An exception of either side stops the other one, is joined and re-thrown on the calling thread.
#pragma once
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>

template <class WaitSlot, class Prepare, class Push, class Pump, class OnAbort>
void run_prep_pipeline(int nchunks, WaitSlot&& wait_slot, Prepare&& prepare, Push&& push, Pump&& pump, OnAbort&& on_abort) {
    struct Pipe {
        std::mutex mu;
        std::condition_variable cv;
        int enqueued = 0;          // chunks whose preparation has been enqueued
        bool abort = false;
        std::exception_ptr err;
    } pipe;
    std::thread producer([&] {
        try {
            for (int k = 0; k < nchunks; ++k) {
                {
                    std::lock_guard<std::mutex> lk(pipe.mu);
                    if (pipe.abort) return;
                }
                wait_slot(k);
                prepare(k);
                {
                    std::lock_guard<std::mutex> lk(pipe.mu);
                    pipe.enqueued = k + 1;
                }
                pipe.cv.notify_all();
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(pipe.mu);
            pipe.err = std::current_exception();
            pipe.cv.notify_all();
        }
    });
    try {
        int next = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(pipe.mu);
                if (pipe.err) std::rethrow_exception(pipe.err);
                while (next < pipe.enqueued) { push(next); ++next; }
            }
            const bool busy = pump();
            if (next == nchunks && !busy) break;
            if (!busy) {           // every lane idle, the next chunk not yet enqueued: wait for the producer
                std::unique_lock<std::mutex> lk(pipe.mu);
                // (system_clock: libstdc++ then waits with pthread_cond_timedwait, which ThreadSanitizer intercepts; the
                // steady-clock form goes through pthread_cond_clockwait, which gcc 11's TSAN does not see - it then believes
                // the mutex stays locked across the wait)
                pipe.cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(2),
                                   [&] { return pipe.enqueued > next || pipe.err; });
            }
        }
    } catch (...) {
        {
            std::lock_guard<std::mutex> lk(pipe.mu);
            pipe.abort = true;
        }
        producer.join();
        on_abort();
        throw;
    }
    producer.join();
}
