// CPU exercise of evidence_amd/csrc/rvll_copypool.h for tests/test_copypool_native.py (ThreadSanitizer / AddressSanitizer
// builds): many copies of every size class through the pool, asleep and polling, with tickets shared between copies, checked
// byte for byte; pools made and torn down in a row; a pool torn down with nothing ever queued.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rvll_copypool.h"

using rvll::CopyPool;

static unsigned long long state = 88172645463325252ull;
static unsigned next_u32() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (unsigned)(state >> 11); }

static int run(int workers, bool polling)
{
    CopyPool pool(workers);
    if (pool.size() != workers) return 1;
    const size_t sizes[] = {1, 7, 4095, 4096, 4097, 128u << 10, (128u << 10) + 1, 1u << 20, (3u << 20) + 12345};
    std::vector<unsigned char> src(8u << 20), dst(8u << 20);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)next_u32();
    if (polling) pool.busy(true);
    for (int round = 0; round < 3; ++round) {
        for (size_t n : sizes) {
            std::fill(dst.begin(), dst.end(), 0);
            const size_t off = next_u32() % (src.size() - n + 1);
            CopyPool::Ticket t;
            // two copies on one ticket, as a chunk's theta rows and log-L go out together
            const size_t half = n / 2;
            pool.copy(dst.data() + off, src.data() + off, half, &t);
            pool.copy(dst.data() + off + half, src.data() + off + half, n - half, &t);
            CopyPool::wait(&t);
            if (std::memcmp(dst.data() + off, src.data() + off, n) != 0) return 2;
            if (off > 0 && dst[off - 1] != 0) return 3;
            if (off + n < dst.size() && dst[off + n] != 0) return 3;
        }
        // several tickets in flight at once
        CopyPool::Ticket ts[4];
        std::fill(dst.begin(), dst.end(), 0);
        for (int k = 0; k < 4; ++k) pool.copy(dst.data() + (size_t)k * (2u << 20), src.data() + (size_t)k * (2u << 20), 2u << 20, &ts[k]);
        for (int k = 3; k >= 0; --k) CopyPool::wait(&ts[k]);
        if (std::memcmp(dst.data(), src.data(), 8u << 20) != 0) return 4;
        if (polling && round == 1) { pool.busy(false); pool.busy(true); }
    }
    if (polling) pool.busy(false);
    return 0;
}

int main()
{
    for (int workers : {1, 4})
        for (bool polling : {false, true}) {
            const int rc = run(workers, polling);
            if (rc) { std::printf("copy pool FAILED: %d workers, polling %d: code %d\n", workers, (int)polling, rc); return 1; }
        }
    for (int i = 0; i < 20; ++i) { CopyPool idle(3); (void)idle; }
    std::printf("copy pool run ok\n");
    return 0;
}
