// exg_filter.hpp — the `filters` predicate of new_reader / exg_open: the text the reference's FilterToString
// renders (exon/src/exon/arrow_table_function/module.cpp:158-214) and DataFusion evaluates as
// `SELECT * FROM exon_table WHERE <filters>` (rust/src/arrow_reader.rs:125-141), parsed into the postfix
// program exg_arrow.hip evaluates on the device.
#pragma once
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <string>
#include <vector>

#include "exg_arrow.hpp"

namespace exg_rd {
namespace ea = exg::arrow;

struct FilterColumn {
    std::string name;
    char kind;  // 'u' Utf8 / VARCHAR, 'l' Int64, 'f' Float32, anything else: not filterable (nested)
};

struct FilterParser {
    const std::string &s;
    size_t i = 0;
    const std::vector<FilterColumn> &cols;
    ea::FilterProgram prog;
    std::string consts;
    std::string err;

    FilterParser(const std::string &text, const std::vector<FilterColumn> &c) : s(text), cols(c) { prog.n_ops = 0; }
    void ws() {
        while (i < s.size() && isspace((unsigned char)s[i])) i++;
    }
    bool kw(const char *k) {
        ws();
        size_t n = strlen(k);
        if (i + n > s.size() || strncasecmp(s.c_str() + i, k, n) != 0) return false;
        if (i + n < s.size() && (isalnum((unsigned char)s[i + n]) || s[i + n] == '_')) return false;
        i += n;
        return true;
    }
    bool push(const ea::FilterOp &op) {
        if (prog.n_ops >= (uint32_t)ea::kMaxFilterOps) return err = "filter too long", false;
        prog.ops[prog.n_ops++] = op;
        return true;
    }
    bool primary() {
        ws();
        if (i < s.size() && s[i] == '(') {
            i++;
            if (!or_expr()) return false;
            ws();
            if (i >= s.size() || s[i] != ')') return err = "expected )", false;
            i++;
            return true;
        }
        std::string name;
        if (i < s.size() && s[i] == '"') {
            i++;
            while (i < s.size() && s[i] != '"') name.push_back(s[i++]);
            i++;
        } else {
            while (i < s.size() && (isalnum((unsigned char)s[i]) || s[i] == '_')) name.push_back(s[i++]);
        }
        if (name.empty()) return err = "expected a column name at '" + s.substr(i) + "'", false;
        int col = -1;
        for (size_t c = 0; c < cols.size(); c++)
            if (strcasecmp(cols[c].name.c_str(), name.c_str()) == 0) col = (int)c;
        if (col < 0) return err = "No field named " + name, false;
        const std::string fmt(1, cols[col].kind);
        if (fmt != "u" && fmt != "l" && fmt != "f") return err = "filters on nested column " + name + " are not supported", false;
        ea::FilterOp op;
        memset(&op, 0, sizeof op);
        op.col = (uint8_t)col;
        if (kw("IS")) {
            bool neg = kw("NOT");
            if (!kw("NULL")) return err = "expected NULL", false;
            op.op = neg ? ea::kOpIsNotNull : ea::kOpIsNull;
            return push(op);
        }
        ws();
        op.op = ea::kOpCmp;
        if (s.compare(i, 2, "!=") == 0 || s.compare(i, 2, "<>") == 0) op.cmp = ea::kNe, i += 2;
        else if (s.compare(i, 2, "<=") == 0) op.cmp = ea::kLe, i += 2;
        else if (s.compare(i, 2, ">=") == 0) op.cmp = ea::kGe, i += 2;
        else if (s.compare(i, 1, "=") == 0) op.cmp = ea::kEq, i += 1;
        else if (s.compare(i, 1, "<") == 0) op.cmp = ea::kLt, i += 1;
        else if (s.compare(i, 1, ">") == 0) op.cmp = ea::kGt, i += 1;
        else return err = "expected a comparison at '" + s.substr(i) + "'", false;
        ws();
        if (i < s.size() && s[i] == '\'') {
            i++;
            std::string lit;
            for (;;) {
                if (i >= s.size()) return err = "unterminated string literal", false;
                if (s[i] == '\'') {
                    if (i + 1 < s.size() && s[i + 1] == '\'') {
                        lit.push_back('\'');
                        i += 2;
                        continue;
                    }
                    i++;
                    break;
                }
                lit.push_back(s[i++]);
            }
            if (fmt != "u") return err = "cannot compare " + name + " with a string", false;
            op.lit = ea::kLitStr;
            op.str_off = (uint32_t)consts.size();
            op.str_len = (uint32_t)lit.size();
            consts += lit;
        } else {
            size_t j = i;
            if (j < s.size() && (s[j] == '-' || s[j] == '+')) j++;
            bool is_float = false;
            while (j < s.size() && (isdigit((unsigned char)s[j]) || s[j] == '.' || s[j] == 'e' || s[j] == 'E' ||
                                    ((s[j] == '-' || s[j] == '+') && (s[j - 1] == 'e' || s[j - 1] == 'E')))) {
                if (!isdigit((unsigned char)s[j])) is_float = true;
                j++;
            }
            if (j == i) return err = "expected a literal at '" + s.substr(i) + "'", false;
            std::string num = s.substr(i, j - i);
            i = j;
            if (fmt == "u") return err = "cannot compare " + name + " with a number", false;
            if (is_float) {
                op.lit = ea::kLitFloat;
                op.f = strtod(num.c_str(), nullptr);
            } else {
                op.lit = ea::kLitInt;
                op.i = strtoll(num.c_str(), nullptr, 10);
                op.f = (double)op.i;
            }
        }
        return push(op);
    }
    bool and_expr() {
        if (!primary()) return false;
        while (kw("AND")) {
            if (!primary()) return false;
            ea::FilterOp op;
            memset(&op, 0, sizeof op);
            op.op = ea::kOpAnd;
            if (!push(op)) return false;
        }
        return true;
    }
    bool or_expr() {
        if (!and_expr()) return false;
        while (kw("OR")) {
            if (!and_expr()) return false;
            ea::FilterOp op;
            memset(&op, 0, sizeof op);
            op.op = ea::kOpOr;
            if (!push(op)) return false;
        }
        return true;
    }
    bool parse() {
        if (!or_expr()) return false;
        ws();
        if (i != s.size()) return err = "unexpected '" + s.substr(i) + "'", false;
        return true;
    }
};


}  // namespace exg_rd
