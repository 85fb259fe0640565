// The handlers that end every extern "C" entry point of the library (no HIP in here: the device-free translation units
// -- formats_api.hip, epilogue_api.hip -- include it too).
#ifndef BLASR_AMD_ABI_GUARD_H_
#define BLASR_AMD_ABI_GUARD_H_
#include <atomic>
#include <exception>
#include <cstdlib>
#include <new>
#include <thread>
#include <vector>

#include "../../include/blasr_hip.h"

// No exception crosses the C ABI: every extern "C" entry point is a function-try-block that ends in one
// of these.  bad_alloc becomes BLASR_HIP_ERR_NOMEM, anything else BLASR_HIP_ERR_INTERNAL; blasr_hip_fail itself cannot
// throw.  Worker threads started inside a call catch in their own body and hand the code to the caller (an exception that
// leaves a std::thread is std::terminate in the middle of the host's process).
#define BLASR_HIP_CATCH_CTX(ctx)                                                                                 \
  catch (const std::bad_alloc &) {                                                                               \
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_NOMEM, "%s: out of host memory", __func__);                         \
  }                                                                                                              \
  catch (const std::exception &e_) {                                                                             \
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INTERNAL, "%s: %s", __func__, e_.what());                           \
  }                                                                                                              \
  catch (...) {                                                                                                  \
    return blasr_hip_fail(ctx, BLASR_HIP_ERR_INTERNAL, "%s: unknown exception", __func__);                       \
  }
#define BLASR_HIP_CATCH_CODE                                    \
  catch (const std::bad_alloc &) { return BLASR_HIP_ERR_NOMEM; } \
  catch (...) { return BLASR_HIP_ERR_INTERNAL; }
#define BLASR_HIP_CATCH_NULL \
  catch (...) { return nullptr; }
// the body of a worker thread: `code` (a std::atomic<int>) receives the first error
#define BLASR_HIP_THREAD_CATCH(code)                                                                        \
  catch (const std::bad_alloc &) { int z_ = 0; (code).compare_exchange_strong(z_, BLASR_HIP_ERR_NOMEM); }   \
  catch (...) { int z_ = 0; (code).compare_exchange_strong(z_, BLASR_HIP_ERR_INTERNAL); }


// How many host threads a pool of the library may use: `want`, cut to BLASR_HIP_HOST_THREADS when the host has set it
// (one process per GPU on an 8-GPU node: bench.py gives every rank cpu_count / world).
inline int blasr_hip_host_threads(int want) {
  if (const char *e = getenv("BLASR_HIP_HOST_THREADS")) {
    const int cap = atoi(e);
    if (cap > 0 && cap < want) want = cap;
  }
  return want < 1 ? 1 : want;
}

// fn(t) for t = 0 .. threads - 1, side by side: the caller's thread takes t = 0 and whatever could not get a thread of its
// own (std::thread's constructor throws when the system has none left).  Nothing thrown inside fn leaves its thread.
// Returns BLASR_HIP_OK, or BLASR_HIP_ERR_NOMEM / BLASR_HIP_ERR_INTERNAL from the first fn that threw.
template <typename F>
int blasr_hip_run_threads(int threads, F fn) {
  std::atomic<int> err(0);
  auto body = [&](int t) {
    try {
      fn(t);
    }
    BLASR_HIP_THREAD_CATCH(err)
  };
  std::vector<std::thread> pool;
  int started = 1;
  try {
    pool.reserve((size_t)(threads > 1 ? threads - 1 : 0));
    for (int t = 1; t < threads; t++) {
      pool.emplace_back(body, t);
      started++;
    }
  } catch (...) {
  }
  body(0);
  for (int t = started; t < threads; t++) body(t);
  for (auto &th : pool) th.join();
  return err.load();
}

#endif
