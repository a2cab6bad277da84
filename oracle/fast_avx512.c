/* CPU ORACLE, TIMED LEG ONLY (test infrastructure): fast_impl.h instantiated for the avx512 backend (see fast_vec.h) */
#define GLV_BACKEND 512
#define FN(name) name##_avx512
#include "fast_impl.h"
