// ThreadSanitizer driver for the staging ring's HOST side (reader thread pool, slot state machine, tickets): a host-memory
// ring (device = -1), several files in flight, chunked reads, slots released and re-submitted immediately, an error
// path (short file) and teardown with work pending.  Built and run by tests/test_ring_sanitizer_cpu.py:
//   clang++ -x hip --offload-host-only -fsanitize=thread ... staging_ring.hip api.hip ring_tsan.cpp -lamdhip64
// Exit code 0 and no "WARNING: ThreadSanitizer" in the output = pass.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/bgamd.h"

static std::string write_file(const std::string& dir, int k, size_t n) {
    std::string p = dir + "/f" + std::to_string(k) + ".bin";
    FILE* f = fopen(p.c_str(), "wb");
    std::vector<unsigned char> buf(n);
    for (size_t i = 0; i < n; ++i) buf[i] = (unsigned char)((i * 31 + k * 7) & 0xff);
    fwrite(buf.data(), 1, n, f);
    fclose(f);
    return p;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t n = 3 * 1024 * 1024 + 123;
    std::vector<std::string> files;
    for (int k = 0; k < 6; ++k) files.push_back(write_file(dir, k, n));
    bg_ring* ring = nullptr;
    if (bg_ring_create(-1, 3, (int64_t)n, 8, &ring) != 0) { fprintf(stderr, "create: %s\n", bg_last_error()); return 2; }
    std::vector<unsigned char> out(n);
    int bad = 0;
    for (int round = 0; round < 4; ++round) {
        int64_t t[3];
        for (int k = 0; k < 3; ++k)
            if (bg_ring_submit(ring, files[(round * 3 + k) % 6].c_str(), 0, (int64_t)n, 5, &t[k]) != 0) { fprintf(stderr, "submit: %s\n", bg_last_error()); return 3; }
        for (int k = 0; k < 3; ++k) {
            if (bg_ring_copy_out(ring, t[k], out.data(), (int64_t)n, nullptr) != 0) { fprintf(stderr, "copy_out: %s\n", bg_last_error()); return 4; }
            const int fk = (round * 3 + k) % 6;
            for (size_t i = 0; i < n; i += 4099) bad += out[i] != (unsigned char)((i * 31 + fk * 7) & 0xff);
            if (bg_ring_release(ring, t[k], nullptr) != 0) return 5;
            // the freed slot is taken again at once, while other slots are still being read
            int64_t t2;
            if (bg_ring_submit(ring, files[fk].c_str(), 100, 4096, 2, &t2) != 0) return 6;
            if (bg_ring_copy_out(ring, t2, out.data(), 4096, nullptr) != 0) return 7;
            bad += out[0] != (unsigned char)((100 * 31 + fk * 7) & 0xff);
            if (bg_ring_release(ring, t2, nullptr) != 0) return 8;
        }
    }
    // error path: a read beyond the end of the file must come back as an error, not hang
    int64_t te;
    if (bg_ring_submit(ring, files[0].c_str(), (int64_t)n - 10, 4096, 3, &te) != 0) return 9;
    void* p = nullptr;
    if (bg_ring_acquire(ring, te, nullptr, &p) == 0) { fprintf(stderr, "short read not reported\n"); return 10; }
    bg_ring_release(ring, te, nullptr);
    // teardown with reads in flight
    int64_t tp[2];
    for (int k = 0; k < 2; ++k) bg_ring_submit(ring, files[k].c_str(), 0, (int64_t)n, 8, &tp[k]);
    bg_ring_destroy(ring);
    if (bad) { fprintf(stderr, "%d corrupted bytes\n", bad); return 11; }
    printf("ring_tsan: ok\n");
    return 0;
}
