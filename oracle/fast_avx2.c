/* CPU ORACLE, TIMED LEG ONLY (test infrastructure): fast_impl.h instantiated for the avx2 backend (see fast_vec.h) */
#define GLV_BACKEND 256
#define FN(name) name##_avx2
#include "fast_impl.h"
