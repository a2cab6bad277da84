/* CPU ORACLE, TIMED LEG ONLY (test infrastructure): fast_impl.h instantiated for the scalar backend (see fast_vec.h) */
#define GLV_BACKEND 1
#define FN(name) name##_scalar
#include "fast_impl.h"
