// Host-side check of key_mix / key_unmix (kmer_device.hpp): for every k the super-k-mer engine takes (20..32) the pair is a
// bijection of the 2k-bit keys -- unmix(mix(x)) == x, mix(x) stays inside 2k bits, distinct keys stay distinct -- and the
// mixed keys of near-copies (keys one base apart) do not share their top digits.  Built and run by tests/test_abi.py with
// hipcc (host code only: no device is touched).
#include <cstdio>
#include <cstdlib>
#include <set>
#include "kmer_device.hpp"

using namespace dnagpu;

int main()
{
    u64 s = 0x1234567;
    int bad = 0;
    for (int k = 20; k <= 32; k++) {
        const u64 mask = kmer_mask(k);
        std::set<u64> seen;
        const u64 edge[] = {0, 1, mask, mask - 1, mask >> 1, (u64)1 << (2 * k - 1), 0x5555555555555555ull & mask};
        for (int i = 0; i < 20007; i++) {
            s = splitmix64(s);
            const u64 x = i < 7 ? edge[i] : (s & mask);
            const u64 y = key_mix(x, k, mask);
            if (y & ~mask) { printf("k=%d: mix(%llx) = %llx leaves the key bits\n", k, (unsigned long long)x, (unsigned long long)y); bad++; }
            if (key_unmix(y, k, mask) != x) { printf("k=%d: unmix(mix(%llx)) = %llx\n", k, (unsigned long long)x, (unsigned long long)key_unmix(y, k, mask)); bad++; }
            if (key_mix(key_unmix(x, k, mask), k, mask) != x) { printf("k=%d: mix(unmix(%llx)) differs\n", k, (unsigned long long)x); bad++; }
            seen.insert(y);
        }
        // near-copies: one key and its 3 k single-base variants -> mostly different top 10 bits
        s = splitmix64(s);
        const u64 base = s & mask;
        std::set<u64> tops;
        int n = 0;
        for (int p = 0; p < k; p++)
            for (u64 d = 1; d < 4; d++) {
                const u64 v = base ^ (d << (2 * p));
                tops.insert(key_mix(v, k, mask) >> (2 * k - 10));
                n++;
            }
        if ((int)tops.size() * 10 < n * 8) { printf("k=%d: %d single-base variants fall into %zu of 1024 top digits\n", k, n, tops.size()); bad++; }
    }
    printf("%s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
