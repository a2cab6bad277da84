// ThreadSanitizer driver for the product's host worker pool (intmax-zkp-core_amd/csrc/host_pool.h through the exports of
// host_selftest.cpp): built and run by tests/test_oracle_sanitizers.py::test_host_worker_pool_under_tsan.
#include <cstdio>
#include <cstdint>
extern "C" uint64_t hs_pool_stress(uint64_t, uint64_t, uint64_t);
extern "C" uint64_t hs_pool_two_contexts(uint64_t, uint64_t);  // two contexts (own thread, own state, own job function) on the one pool
int main() {
    uint64_t bad = hs_pool_stress(1, 300, 64) + hs_pool_stress(3, 100, 257) + hs_pool_stress(2, 5, 20000) +
                   hs_pool_two_contexts(200, 64) + hs_pool_two_contexts(20, 4000);
    printf("bad=%llu\n", (unsigned long long)bad);
    return bad != 0;
}
