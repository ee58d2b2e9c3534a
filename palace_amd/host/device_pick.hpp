// Which GPU an executable works on.  The reference's stages are one process each and know nothing of devices; a node with eight
// MI355X runs eight samples side by side with PALACE_DEVICE=<k> per pipeline (the `weak` reading of "N GPUs": no collective in the
// data path).  Default: device 0.  When nothing else restricts the visible devices, the choice is made by ROCR_VISIBLE_DEVICES,
// set here BEFORE the process's first HIP call: the runtime then brings up one device instead of every device of the node (its
// start-up is most of eref's wall time), and the chosen device is ordinal 0 of the process.
#pragma once
#include <cstdlib>
#include <string>

#include "fast_exit.hpp"

namespace palace_host {

// call once, from main(), before any thread that may touch HIP exists; returns the ordinal to hand to palace_ctx_create
inline int pick_device()
{
    static const int chosen = [] {
        const char* d = std::getenv("PALACE_DEVICE");
        int dev = (d && *d) ? std::atoi(d) : 0;
        if (dev < 0) dev = 0;
        for (const char* k : {"ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"})
            if (std::getenv(k)) return dev;                       // somebody has chosen the visible set: PALACE_DEVICE is an ordinal within it
        if (gpu_touched_before_main()) return dev;               // (a runtime is up already: the variable would come too late)
        ::setenv("ROCR_VISIBLE_DEVICES", std::to_string(dev).c_str(), 1);
        return 0;
    }();
    return chosen;
}

}  // namespace palace_host
