// Small-block free lists for the host side's handle churn (late round 5).
//
// An end-to-end program is 10^5-10^6 API calls, each of which creates and soon drops a few small objects — the handle, its
// buffer descriptor, a deferred chain, the interpreter's wrapper, DAG nodes, cache entries.  The interpreter descends through
// a nest of `if`s allocating thousands of them and frees them on the way back up, and glibc's per-thread cache holds 7 blocks
// per size: everything beyond goes through its slow paths.  A PC profile of the calling thread on mixture (whose wall clock IS
// that thread, profiles/r05/host_threads_split.txt) had 40 % of its samples inside malloc / free, and raising glibc's
// tcache_count — a start-up tunable, not something a library can set — took the program from 0.120-0.143 to 0.100-0.118 s
// (profiles/r05/host_profile.txt).  So the hot fixed-size objects come from per-thread LIFO free lists by 16-byte size class,
// filled by what the thread itself frees; a list holds at most LIST_BYTES, the rest goes back to operator delete, and a thread's
// lists are released when it ends.  Blocks freed on another thread than the one that allocated them simply join that thread's
// lists (they are plain operator-new blocks of the class size).
#pragma once
#include <cstddef>
#include <cstdlib>
#include <new>
#include <type_traits>

namespace gft_small {

constexpr size_t MAX_BYTES = 4096;          // larger requests: operator new / delete (round 6: the launch graph's recordings are 1-3 KB)
constexpr size_t LIST_BYTES = 64u << 20;    // per size class and thread (a program's pending recordings: tens of thousands of 1-3 KB blocks)
constexpr size_t NCLASS = MAX_BYTES / 16;

// ONE constant-initialised thread_local per thread (a shared library reaches a thread_local through a __tls_get_addr call, and
// one with a constructor through a guard test on top: 4 % of mixture's calling thread when `dead` and the lists were two of them):
//   dead   the thread is ending (or the process is, for the main thread): straight to operator new / delete.  Later thread_local
//          / static destructors that release handles read it after the lists have been destroyed;
//   lists  this thread's free lists once it has used them (owned by the thread_local in lists_slow()).
struct Lists;
struct Tls {
    bool dead;
    Lists* lists;
};
inline Tls& tls() {
    static thread_local Tls t = {false, nullptr};
    return t;
}
inline bool& dead() { return tls().dead; }
struct Lists {
    void* head[NCLASS + 1] = {};
    unsigned count[NCLASS + 1] = {};
    ~Lists() {
        Tls& t = tls();
        t.dead = true;
        t.lists = nullptr;
        for (size_t c = 0; c <= NCLASS; ++c) {
            while (void* p = head[c]) {
                head[c] = *static_cast<void**>(p);
                ::operator delete(p);
            }
            count[c] = 0;
        }
    }
};
inline Lists& lists_slow() {  // first use on this thread
    static thread_local Lists L;
    tls().lists = &L;
    return L;
}
inline Lists& lists() { return tls().lists ? *tls().lists : lists_slow(); }
inline void* get(size_t bytes) {
    if (bytes == 0) bytes = 1;
    if (bytes > MAX_BYTES) return ::operator new(bytes);
    const size_t c = (bytes + 15) >> 4;
    Tls& t = tls();
    if (t.dead) return ::operator new(c << 4);
    Lists& L = t.lists ? *t.lists : lists_slow();
    if (void* p = L.head[c]) {
        L.head[c] = *static_cast<void**>(p);
        --L.count[c];
        return p;
    }
    return ::operator new(c << 4);
}
inline void put(void* p, size_t bytes) noexcept {
    if (!p) return;
    if (bytes == 0) bytes = 1;
    if (bytes > MAX_BYTES) {
        ::operator delete(p);
        return;
    }
    const size_t c = (bytes + 15) >> 4;
    Tls& t = tls();
    if (t.dead) {
        ::operator delete(p);
        return;
    }
    Lists& L = t.lists ? *t.lists : lists_slow();
    if ((size_t)L.count[c] * (c << 4) >= LIST_BYTES) {
        ::operator delete(p);
        return;
    }
    *static_cast<void**>(p) = L.head[c];
    L.head[c] = p;
    ++L.count[c];
}

// std allocator over the lists (std::allocate_shared, node containers, small vectors)
template <class T>
struct Alloc {
    typedef T value_type;
    Alloc() noexcept {}
    template <class U>
    Alloc(const Alloc<U>&) noexcept {}
    T* allocate(size_t n) {
        static_assert(alignof(T) <= 16, "the lists hand out 16-byte aligned blocks");
        return static_cast<T*>(get(n * sizeof(T)));
    }
    void deallocate(T* p, size_t n) noexcept { put(const_cast<typename std::remove_const<T>::type*>(p), n * sizeof(T)); }
    template <class U>
    bool operator==(const Alloc<U>&) const noexcept { return true; }
    template <class U>
    bool operator!=(const Alloc<U>&) const noexcept { return false; }
};

}  // namespace gft_small
