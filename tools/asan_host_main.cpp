#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
extern "C" int gfh_run(const char*, const char*, const char*, const char*, char**, char**);
extern "C" void gfh_free(void*);
int main(int argc, char** argv) {
    int bad = 0;
    for (int i = 3; i < argc; ++i) {
        std::ifstream f(argv[i]);
        std::stringstream ss;
        ss << f.rdbuf();
        std::string src = ss.str(), flags;
        if (src.rfind("# flags:", 0) == 0) flags = src.substr(8, src.find('\n') - 8);
        flags += " --no-timing";
        char *out = nullptr, *tj = nullptr;
        int rc = gfh_run(src.c_str(), flags.c_str(), argv[1], argv[2], &out, &tj);
        if (rc != 0) { bad++; std::printf("rc=%d %s\n", rc, argv[i]); }
        if (out) gfh_free(out);
        if (tj) gfh_free(tj);
    }
    std::printf("done, %d nonzero rc\n", bad);
    return 0;
}
