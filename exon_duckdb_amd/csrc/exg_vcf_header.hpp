// exg_vcf_header.hpp — host only: the typed keys a VCF header declares (`##INFO=<ID=..,Number=..,Type=..>` / `##FORMAT=<..>`),
// which are the children of the `info` STRUCT and of the `formats` LIST(STRUCT) (exon 0.2.6 VCFSchemaBuilder over
// noodles-vcf 0.34 Header::infos / ::formats).  Everything read here is a user's file: the TU is built with
// ASan / UBSan in tests/host_asan_driver.cpp.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace exg_rd {

// value types of a key (the same numbers as exg::arrow::kVt* in exg_arrow.hpp)
enum : uint8_t { kKeyFlag = 0, kKeyInt = 1, kKeyFloat = 2, kKeyString = 3 };

struct KeyDef {
    std::string id;
    uint8_t type = kKeyString;
    bool is_list = false;  // Number != 1 (and not a Flag): a LIST of its type
};

// the keys of header text d[0, n) (the leading '#' lines), in header order, first definition of an ID wins
void parse_vcf_header(const char *d, size_t n, std::vector<KeyDef> *info, std::vector<KeyDef> *format);

// "INFO DP:i AF:[f] DB:b ANN:u | FORMAT GT:u AD:[i]" (what the CPU tests compare)
std::string explain_vcf_header(const char *d, size_t n);

}  // namespace exg_rd
