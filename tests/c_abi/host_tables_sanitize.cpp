// Host-only driver for the sanitizer builds (tests/test_host_logic.py): every table builder of nus_tables.cpp over a sweep of
// sizes, ratios and filters -- including the degenerate ones (1-pixel axes, windows wider than the table) -- the exact-ratio
// phase-frame views, the down-scaling stream tables, serialisation round trips and DAMAGED blobs (truncated at every length
// class, every header word overwritten), and the frame queue under a producer and two consumers.  Nothing is asserted about
// values here (tests/test_host_logic.py compares them with the oracle's bit for bit); the point is that -fsanitize=address,
// undefined sees every index these functions form.
#include "nus_queue.hpp"
#include "nus_tables.hpp"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>

using namespace nus;

int main()
{
    int bad = 0;
    const uint32_t ins[] = {1, 2, 3, 4, 5, 7, 8, 16, 17, 63, 64, 240, 321, 1080, 1920};
    const double ratios[] = {0.2, 0.25, 1.0 / 3, 0.5, 0.75, 1.0, 4.0 / 3, 1.5, 5.0 / 3, 2.0, 2.5, 3.0, 4.0, 6.0};
    size_t built = 0;
    for (uint32_t in_n : ins)
        for (double r : ratios) {
            uint32_t out_n = (uint32_t)(in_n * r + 0.5);
            if (out_n == 0) out_n = 1;
            for (int f = 0; f < 3; ++f) {
                AxisTables x, y;
                build_axis_tables(in_n, out_n, f == 1, x, (ResizeFilter)f);
                build_axis_tables(in_n, out_n, false, y, (ResizeFilter)f);
                ++built;
                std::vector<float> w6;
                std::vector<uint32_t> cls;
                std::vector<float> classes;
                if (x.lz_max_taps > 0) {
                    if (out_n == 2 * in_n && lanczos_x2_phase_frame(x, w6)) (void)lanczos_x2_interior_uniform(x, w6);
                    for (uint32_t S : {3u, 4u})
                        if (out_n == S * in_n && lanczos_xs_phase_frame(x, S, w6)) {
                            (void)lanczos_xs_interior_uniform(x, S, w6);
                            (void)lanczos_xs_weight_classes(x, S, w6, true, cls, classes);
                            (void)lanczos_xs_weight_classes(x, S, w6, false, cls, classes);
                        }
                    if (2 * out_n == 3 * in_n && in_n % 2 == 0 && lanczos_r32_phase_frame(x, w6)) {
                        (void)lanczos_r32_weight_classes(x, w6, true, cls, classes);
                        (void)lanczos_r32_weight_classes(x, w6, false, cls, classes);
                    }
                    if (3 * out_n == 4 * in_n && lanczos_r43_phase_frame(x, w6)) (void)lanczos_r43_interior_uniform(x, w6);
                    if (out_n < in_n) {
                        std::vector<uint32_t> rows;
                        std::vector<int32_t> done;
                        (void)build_down_stream_tables(x, rows, done);
                    }
                }
                // round trip, then damage
                const std::vector<uint8_t> blob = serialize_tables(x, y);
                AxisTables x2, y2;
                std::string err;
                if (!deserialize_tables(blob.data(), blob.size(), x2, y2, err)) ++bad;
                if (x2.nn_src != x.nn_src || x2.lz_w != x.lz_w || y2.bl_frac != y.bl_frac) ++bad;
                if (in_n <= 64) {
                    for (size_t cut : {(size_t)0, (size_t)1, (size_t)7, (size_t)8, (size_t)31, blob.size() / 2, blob.size() - 1}) {
                        if (cut >= blob.size()) continue;
                        std::vector<uint8_t> part(blob.begin(), blob.begin() + cut); // its own allocation: an over-read is seen
                        AxisTables a, b;
                        if (deserialize_tables(part.data(), part.size(), a, b, err)) ++bad;
                    }
                    for (size_t word = 0; word + 4 <= blob.size() && word < 96; word += 4)
                        for (uint32_t v : {0u, 1u, 0x7FFFFFFFu, 0xFFFFFFFFu, 0x10000u}) {
                            std::vector<uint8_t> dmg(blob);
                            memcpy(&dmg[word], &v, 4);
                            AxisTables a, b;
                            (void)deserialize_tables(dmg.data(), dmg.size(), a, b, err); // may succeed or fail; must not overrun
                        }
                }
            }
        }
    // frame queue: one producer, two consumers, drop-oldest
    {
        FrameQueue q(5);
        std::atomic<bool> stop{false};
        std::atomic<uint64_t> got{0};
        std::thread prod([&] {
            std::vector<uint8_t> px(64 * 48 * 4);
            for (int i = 0; i < 4000; ++i) {
                px[0] = (uint8_t)i;
                q.add(px.data(), 64, 48);
            }
            stop = true;
        });
        auto cons = [&] {
            while (!stop || q.size()) {
                bool too_big = false;
                if (auto f = q.pop(1, 64 * 48 * 4, &too_big)) got += f->data.size() == 64 * 48 * 4;
                (void)q.latest(0);
            }
        };
        std::thread c1(cons), c2(cons);
        prod.join();
        c1.join();
        c2.join();
        if (got + q.dropped() != 4000) ++bad;
    }
    printf("tables %zu bad %d\n", built, bad);
    return bad ? 1 : 0;
}
