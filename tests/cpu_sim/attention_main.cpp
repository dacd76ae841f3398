// host run of mv_attention_f16 / mv_temporal_attention_f16 (real dispatch + kernels of musev_amd/csrc/attention.hip)
//   argv: dir     dir/job.txt "key value" lines; dir/q.bin, k<i>.bin, v<i>.bin (fp16), out0.bin (initial out when accumulating)
#include "attention_sim.inc"

#include <map>
#include <string>

static thread_local char g_err[512] = "";
void mv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::vector<char> slurp(const std::string& p) {
    std::vector<char> v;
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n + 256);
    if (fread(v.data(), 1, n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    const std::string dir = argv[1];
    std::map<std::string, double> kv;
    {
        FILE* f = fopen((dir + "/job.txt").c_str(), "r");
        if (!f) return 3;
        char key[64];
        double val;
        while (fscanf(f, "%63s %lf", key, &val) == 2) kv[key] = val;
        fclose(f);
    }
    auto q = slurp(dir + "/q.bin");
    const long rows = (long)kv["rows"], ldo = (long)kv["ldo"];
    std::vector<char> out = slurp(dir + "/out0.bin");
    if (out.empty()) out.assign((size_t)rows * ldo * 2 + 256, 0);
    int rc;
    std::vector<std::vector<char>> keep;
    if (kv["temporal"] != 0) {
        auto k = slurp(dir + "/k0.bin"), v = slurp(dir + "/v0.bin");
        rc = mv_temporal_attention_f16(q.data(), k.data(), v.data(), (int)kv["ldq"], (int)kv["ldk0"], (int)kv["ldv0"], out.data(), (int)ldo,
                                       (int)kv["b"], (int)kv["t"], (int)kv["hw"], (int)kv["heads"], (int)kv["d"], (float)kv["scale"], nullptr);
    } else {
        mv_attn_desc d;
        memset(&d, 0, sizeof(d));
        d.q = q.data(); d.out = out.data(); d.ldq = (int)kv["ldq"]; d.ldo = (int)ldo;
        d.nb = (int)kv["nb"]; d.lq = (int)kv["lq"]; d.heads = (int)kv["heads"]; d.d = (int)kv["d"];
        d.scale = (float)kv["scale"]; d.nseg = (int)kv["nseg"]; d.accumulate = (int)kv["accumulate"]; d.out_scale = (float)kv["out_scale"];
        for (int s = 0; s < d.nseg; ++s) {
            const std::string i = std::to_string(s);
            keep.push_back(slurp(dir + "/k" + i + ".bin"));
            d.seg[s].k = keep.back().data();
            keep.push_back(slurp(dir + "/v" + i + ".bin"));
            d.seg[s].v = keep.back().data();
            d.seg[s].ldk = (int)kv["ldk" + i]; d.seg[s].ldv = (int)kv["ldv" + i]; d.seg[s].len = (int)kv["len" + i];
            d.seg[s].div = (int)kv["div" + i]; d.seg[s].mul = (int)kv["mul" + i]; d.seg[s].add = (int)kv["add" + i];
        }
        rc = mv_attention_f16(&d, nullptr);
    }
    if (rc != 0) {
        fprintf(stderr, "attention failed (%d): %s\n", rc, g_err);
        return 5;
    }
    FILE* f = fopen((dir + "/out.bin").c_str(), "wb");
    fwrite(out.data(), 2, (size_t)rows * ldo, f);
    fclose(f);
    return 0;
}
