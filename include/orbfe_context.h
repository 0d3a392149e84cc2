// orbfe_context.h -- one liborbfe matcher context (HIP stream + pinned / device staging blocks) per host thread.
//
// ORB_SLAM2::ORBmatcher objects are stack temporaries created at every call site and used from the Tracking, LocalMapping
// and LoopClosing threads at once (reference include/ORBmatcher.h:100-101, src/Tracking.cc:765, src/LocalMapping.cc:215,
// src/LoopClosing.cc:239); a stream per temporary would cost more than the searches.  The adapters therefore borrow the
// calling thread's context: created on first use, destroyed with the thread.  Device: environment ORBFE_DEVICE (default 0),
// the same rule as adapters/ORBextractor_orbfe.cc.
#pragma once
#include <cstdlib>

#include "orbfe.h"

namespace orbfe {

inline int default_device() {
    static const int dev = [] { const char *d = std::getenv("ORBFE_DEVICE"); return d ? std::atoi(d) : 0; }();
    return dev;
}

// the calling thread's context, or nullptr with *status set (orbfe_last_error() has the message)
inline orbfe_matcher *thread_matcher(int *status = nullptr) {
    struct Holder {
        orbfe_matcher *m = nullptr;
        ~Holder() { orbfe_matcher_destroy(m); }
    };
    static thread_local Holder h;
    int rc = ORBFE_OK;
    if (!h.m) rc = orbfe_matcher_create(default_device(), &h.m);   // retried on every call until a device is there
    if (status) *status = rc;
    return h.m;
}

}  // namespace orbfe
