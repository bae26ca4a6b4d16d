// cuda_utils.h counterpart: the reference wraps every runtime call in checkCudaErrors, which
// prints "CUDA ERROR = <code> at <file>:<line> '<expr>'", resets the device and exit(99)s
// (cuda_utils.h:6-16).  Same behaviour on top of the C ABI's status codes.
#pragma once

#include <cstdlib>
#include <iostream>

#include "../../../include/pt_amd.h"

#define checkPtErrors(val) ::ptamd::check_pt((val), #val, __FILE__, __LINE__)

namespace ptamd {
inline void check_pt(pt_status result, const char *func, const char *file, int line)
{
    if (result != PT_OK) {
        std::cerr << "HIP ERROR = " << static_cast<unsigned int>(result) << " at " << file << ":" << line << " '" << func
                  << "' : " << pt_last_error_string() << " \n";
        std::exit(99);
    }
}
} // namespace ptamd
