// CPU check of genfer_amd/csrc/gft_small_alloc.hpp (built and run by tests/test_small_alloc.py under AddressSanitizer):
// size classes, reuse, the per-class bound, blocks freed on another thread, containers over the allocator, thread exit.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../genfer_amd/csrc/gft_small_alloc.hpp"

struct Rec {
    double v[7];
    std::shared_ptr<int> keep;
};

int main() {
    using namespace gft_small;
    // 1. a freed block is the next one handed out for its class; other classes do not see it
    void* a = get(40);
    std::memset(a, 0xab, 40);
    put(a, 40);
    void* b = get(33);  // same 48-byte class
    assert(b == a);
    void* c = get(64);
    assert(c != a);
    put(b, 33);
    put(c, 64);
    // 2. large requests bypass the lists
    void* big = get(MAX_BYTES + 1);
    put(big, MAX_BYTES + 1);
    assert(lists().count[NCLASS] <= LIST_BYTES / MAX_BYTES);
    // 3. the per-class bound: more blocks than LIST_BYTES holds go back to operator delete
    {
        const size_t n = LIST_BYTES / 512 + 1000;
        std::vector<void*> blocks(n);
        for (auto& p : blocks) p = get(512);
        for (auto p : blocks) put(p, 512);
        assert((size_t)lists().count[NCLASS] * 512 <= LIST_BYTES);
    }
    // 4. allocate_shared / node containers / vectors over Alloc
    {
        std::vector<std::shared_ptr<Rec>> v;
        for (int i = 0; i < 10000; ++i) {
            auto r = std::allocate_shared<Rec>(Alloc<Rec>());
            r->keep = std::allocate_shared<int>(Alloc<int>(), i);
            v.push_back(r);
        }
        long sum = 0;
        for (auto& r : v) sum += *r->keep;
        assert(sum == 10000L * 9999 / 2);
        std::unordered_map<int, Rec, std::hash<int>, std::equal_to<int>, Alloc<std::pair<const int, Rec>>> m;
        for (int i = 0; i < 5000; ++i) m[i].v[0] = i;
        for (int i = 0; i < 5000; i += 2) m.erase(i);
        assert(m.size() == 2500 && m[4999].v[0] == 4999);
        std::vector<double, Alloc<double>> small(5, 1.0), copy = small;
        copy.push_back(2.0);
        assert(copy.size() == 6 && small.size() == 5);
    }
    // 5. blocks allocated here and freed on another thread (the launch worker drops closures that hold descriptors), and a
    //    thread's lists released when it ends
    {
        std::vector<std::shared_ptr<Rec>> v;
        for (int i = 0; i < 20000; ++i) v.push_back(std::allocate_shared<Rec>(Alloc<Rec>()));
        std::thread t([&] {
            v.clear();
            auto r = std::allocate_shared<Rec>(Alloc<Rec>());  // reuses one of them on this thread
            r->v[0] = 1.0;
        });
        t.join();
        for (int i = 0; i < 20000; ++i) v.push_back(std::allocate_shared<Rec>(Alloc<Rec>()));
    }
    std::puts("small_alloc ok");
    return 0;
}
