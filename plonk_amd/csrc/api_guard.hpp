// The exception barrier of the C-ABI (HIP-free: also compiled by the host test harness, tests/csrc/host_arith.cpp).
#pragma once
#include <exception>
#include <new>

#include "../../include/plonk_hip.h"

namespace plonk {

void set_last_error(const char* what, const char* detail, const char* file, int line);   // capi.hip; never allocates

// Every int-returning entry point of the C-ABI runs its body inside this guard: a C++ exception (std::bad_alloc from a host
// staging vector, std::system_error from a mutex) becomes an error code instead of unwinding into a C / Rust caller
// ("nothing throws or aborts across the ABI", include/plonk_hip.h).  set_last_error itself does not allocate.
template <class F>
static inline int api_guard(const char* fn, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    set_last_error(fn, "out of host memory (std::bad_alloc)", __FILE__, __LINE__);
    return PLONK_ERR_NOMEM;
  } catch (const std::exception& e) {
    set_last_error(fn, e.what(), __FILE__, __LINE__);
    return PLONK_ERR_STATE;
  } catch (...) {
    set_last_error(fn, "unknown C++ exception", __FILE__, __LINE__);
    return PLONK_ERR_STATE;
  }
}

}  // namespace plonk
