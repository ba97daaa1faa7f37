// stand-in for <tbb/concurrent_vector.h> — included by the reference's map_eval.h, never used.  Test infrastructure only.
#pragma once
