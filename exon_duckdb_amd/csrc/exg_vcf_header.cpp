// exg_vcf_header.cpp — see exg_vcf_header.hpp (host only, no HIP call).
#include "exg_vcf_header.hpp"

#include <string.h>

namespace exg_rd {

// `##INFO=<ID=DP,Number=1,Type=Integer,...>` / `##FORMAT=<...>` in header order (noodles-vcf Header::infos /
// ::formats are insertion-ordered maps)
void parse_vcf_header(const char *d, size_t n, std::vector<KeyDef> *info, std::vector<KeyDef> *format) {
    size_t pos = 0;
    while (pos < n && d[pos] == '#') {
        const char *nl = (const char *)memchr(d + pos, '\n', n - pos);
        size_t end = nl ? (size_t)(nl - d) : n;
        std::string line(d + pos, end - pos);
        pos = nl ? end + 1 : n;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::vector<KeyDef> *dst = nullptr;
        size_t lt = 0;
        if (line.compare(0, 8, "##INFO=<") == 0) dst = info, lt = 8;
        if (line.compare(0, 10, "##FORMAT=<") == 0) dst = format, lt = 10;
        if (!dst) continue;
        // key=value pairs separated by ',' outside double quotes
        KeyDef k;
        std::string number = "1", type = "String";
        size_t i = lt;
        while (i < line.size() && line[i] != '>') {
            size_t eq = line.find('=', i);
            if (eq == std::string::npos) break;
            std::string key = line.substr(i, eq - i), val;
            size_t j = eq + 1;
            if (j < line.size() && line[j] == '"') {
                j++;
                while (j < line.size() && line[j] != '"') {
                    if (line[j] == '\\' && j + 1 < line.size()) j++;
                    val.push_back(line[j++]);
                }
                j++;
            } else {
                while (j < line.size() && line[j] != ',' && line[j] != '>') val.push_back(line[j++]);
            }
            if (key == "ID") k.id = val;
            if (key == "Number") number = val;
            if (key == "Type") type = val;
            i = j < line.size() && line[j] == ',' ? j + 1 : j;
        }
        if (k.id.empty()) continue;
        k.type = type == "Integer" ? kKeyInt : type == "Float" ? kKeyFloat : type == "Flag" ? kKeyFlag : kKeyString;
        k.is_list = k.type != kKeyFlag && number != "1";
        bool dup = false;
        for (auto &o : *dst) dup = dup || o.id == k.id;
        if (!dup) dst->push_back(k);
    }
}

std::string explain_vcf_header(const char *d, size_t n) {
    std::vector<KeyDef> info, format;
    parse_vcf_header(d, n, &info, &format);
    auto one = [](const KeyDef &k) {
        const char *t = k.type == kKeyInt ? "i" : k.type == kKeyFloat ? "f" : k.type == kKeyFlag ? "b" : "u";
        return k.id + ":" + (k.is_list ? std::string("[") + t + "]" : std::string(t));
    };
    std::string res = "INFO";
    for (auto &k : info) res += " " + one(k);
    res += " | FORMAT";
    for (auto &k : format) res += " " + one(k);
    return res;
}

}  // namespace exg_rd
