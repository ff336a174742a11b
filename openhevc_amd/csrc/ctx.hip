// ctx.hip -- host side of libohevc_hip.so: picture store, job recorder, phase-ordered executor (include/ohevc_ctx.h).
// Host code only; the kernels it launches live in the *_kernels.hip files and are reached through the same
// ohevc_dev_* entry points an external caller would use.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <atomic>
#include <string.h>
#include <vector>
#include "common.hpp"
#include "ohevc_ctx.h"
#include "ohevc_debug.h"

// The host layer is ONE translation unit (file-scope state and helpers are shared from top to bottom) in six parts, in dependency order
// ("split ctx.hip (store / recorder / executor / frame end) so that the frame end can be reviewed"):
#include "ctx_store.hpp"         // the device picture store, the context and its life cycle
#include "ctx_pictures.hpp"      // pictures: allocation, upload / copy-back, page-locked host memory, export / import, up-sampling
#include "ctx_record.hpp"        // the recorder: ohevc_frame_begin, ohevc_rec_*
#include "ctx_exec.hpp"          // the executor: upload, ordering, ohevc_frame_reconstruct
#include "ctx_frame_end.hpp"     // the frame end, asynchronous / parked frame ends
#include "ctx_debug.hpp"         // inspection entry points, statistics
