// `genfer` — the reference's command line (src/main.rs:22-131) over the host interpreter and the MI355X Taylor
// core:   genfer [flags] <file.sgcl>
// Same flags (f64 and `--bounds` Taylor paths), same report on stdout, same `--json <path>` file, so the reference's
// own harnesses (benchmarks/neurips2023/exact/bench.py:44-105 spawns `genfer <flags> <path>` and parses
// "Total inference time") can drive it unchanged.  The TaylorPoly backend is libgftaylor.so (HIP, gfx950) next to
// this binary's directory; GENFER_BACKEND=<lib>[:prefix] selects another library exporting the C ABI of
// include/gftaylor.h.
#include <libgen.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>

extern "C" {
int gfh_run(const char* source, const char* flags, const char* backend_lib, const char* backend_prefix, char** out_text,
            char** out_timings_json);
const char* gfh_last_stderr();
void gfh_free(void* p);
}

int main(int argc, char** argv) {
    // This executable owns its process: kernel arguments in device memory shorten every launch of a launch-bound
    // program (libgftaylor documents the flag; the library itself never touches the environment).  A value the user
    // exported wins.  Must happen before the backend library brings up HIP.
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
    std::string file, flags;
    bool bounds = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        const bool takes_value = a == "-l" || a == "--limit" || a == "-u" || a == "--unroll" || a == "--json" || a == "-p" || a == "--precision";
        if (a == "-b" || a == "--bounds") bounds = true;
        if (!a.empty() && a[0] == '-') {
            flags += a + " ";
            if (takes_value && i + 1 < argc) flags += std::string(argv[++i]) + " ";
        } else if (file.empty()) {
            file = a;
        } else {
            fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str());
            return 2;
        }
    }
    if (file.empty()) {
        fprintf(stderr, "Usage: genfer [OPTIONS] <FILE_NAME>\n");
        return 2;
    }
    std::ifstream in(file);
    if (!in) {
        fprintf(stderr, "error: cannot read %s\n", file.c_str());
        return 2;
    }
    std::stringstream src;
    src << in.rdbuf();
    // model name = file stem (main.rs:603)
    std::string stem = file.substr(file.find_last_of('/') == std::string::npos ? 0 : file.find_last_of('/') + 1);
    if (stem.find_last_of('.') != std::string::npos && stem.find_last_of('.') > 0) stem = stem.substr(0, stem.find_last_of('.'));
    flags += "--model-name " + stem + " ";

    std::string lib, prefix = bounds ? "gfti_" : "gft_";
    if (const char* e = getenv("GENFER_BACKEND")) {
        lib = e;
        size_t c = lib.find(':');
        if (c != std::string::npos) {
            std::string pfx = lib.substr(c + 1);
            lib = lib.substr(0, c);
            prefix = bounds ? pfx + "i_" : pfx + "_";  // e.g. "orc" -> orc_ / orci_
        }
    } else {
        char exe[4096];
        ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
        if (n <= 0) {
            fprintf(stderr, "error: cannot locate the executable\n");
            return 2;
        }
        exe[n] = 0;
        lib = std::string(dirname(exe)) + "/../libgftaylor.so";
    }
    char *text = nullptr, *timings = nullptr;
    int rc = gfh_run(src.str().c_str(), flags.c_str(), lib.c_str(), prefix.c_str(), &text, &timings);
    if (text) fputs(text, rc == 0 ? stdout : stderr);
    fputs(gfh_last_stderr(), stderr);
    gfh_free(text);
    gfh_free(timings);
    return rc == 0 ? 0 : 101;  // a Rust panic exits with 101
}
