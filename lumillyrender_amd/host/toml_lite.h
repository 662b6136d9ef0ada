// toml_lite.h -- small TOML reader for scene files (stands in for the `toml` crate the reference
// uses at description.rs:38).  Supports what scene descriptions need: tables, dotted headers,
// arrays of tables ([[object]], [[object.transform]]), inline tables and arrays, basic/literal
// strings, integers, floats, booleans, comments.  Dates and multi-line strings are rejected.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace lrtoml {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
  enum Kind { STRING, INT, FLOAT, BOOL, ARRAY, TABLE } kind = TABLE;
  std::string s;
  int64_t i = 0;
  double f = 0.0;
  bool b = false;
  std::vector<ValuePtr> arr;
  std::map<std::string, ValuePtr> tab;
  std::vector<std::string> key_order;   // insertion order of tab
  bool is_table_array = false;          // ARRAY created by [[header]]
  bool inline_closed = false;           // inline tables / arrays may not be extended by headers

  const Value* get(const std::string& k) const {
    auto it = tab.find(k);
    return it == tab.end() ? nullptr : it->second.get();
  }
  bool is_number() const { return kind == INT || kind == FLOAT; }
  double number() const { return kind == INT ? (double)i : f; }
};

struct ParseError : std::runtime_error {
  ParseError(const std::string& m, int line) : std::runtime_error("toml: line " + std::to_string(line) + ": " + m) {}
};

class Parser {
 public:
  explicit Parser(const std::string& text) : t_(text) {}

  ValuePtr parse() {
    root_ = std::make_shared<Value>();
    root_->kind = Value::TABLE;
    Value* cur = root_.get();
    while (true) {
      skip_ws_nl();
      if (eof()) break;
      if (peek() == '[') {
        cur = parse_header();
      } else {
        parse_keyval(cur);
      }
      skip_ws();
      skip_comment();
      if (!eof() && peek() != '\n' && peek() != '\r') fail("expected end of line");
    }
    return root_;
  }

 private:
  const std::string& t_;
  size_t p_ = 0;
  int line_ = 1;
  int depth_ = 0;                      // nesting of arrays / inline tables being parsed (parse_value recurses: bounded, an input cannot size the stack)
  ValuePtr root_;

  bool eof() const { return p_ >= t_.size(); }
  char peek(size_t o = 0) const { return p_ + o < t_.size() ? t_[p_ + o] : '\0'; }
  char next() { char c = t_[p_++]; if (c == '\n') ++line_; return c; }
  [[noreturn]] void fail(const std::string& m) const { throw ParseError(m, line_); }

  void skip_ws() { while (!eof() && (peek() == ' ' || peek() == '\t')) ++p_; }
  void skip_comment() { if (!eof() && peek() == '#') while (!eof() && peek() != '\n') ++p_; }
  void skip_ws_nl() {
    while (!eof()) {
      char c = peek();
      if (c == ' ' || c == '\t' || c == '\r') ++p_;
      else if (c == '\n') next();
      else if (c == '#') skip_comment();
      else break;
    }
  }
  static bool bare_char(char c) {
    return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_' || c == '-';
  }

  std::string parse_basic_string() {
    // opening quote consumed by caller check
    if (peek() == '"' && peek(1) == '"' && peek(2) == '"') fail("multi-line strings are not supported");
    next();
    std::string out;
    while (true) {
      if (eof() || peek() == '\n') fail("unterminated string");
      char c = next();
      if (c == '"') break;
      if (c == '\\') {
        if (eof()) fail("bad escape");
        char e = next();
        switch (e) {
          case 'b': out += '\b'; break; case 't': out += '\t'; break; case 'n': out += '\n'; break;
          case 'f': out += '\f'; break; case 'r': out += '\r'; break; case '"': out += '"'; break;
          case '\\': out += '\\'; break;
          case 'u': case 'U': {
            int n = e == 'u' ? 4 : 8; uint32_t cp = 0;
            for (int k = 0; k < n; ++k) {
              char h = eof() ? '\0' : next(); int d;
              if (h >= '0' && h <= '9') d = h - '0'; else if (h >= 'a' && h <= 'f') d = h - 'a' + 10;
              else if (h >= 'A' && h <= 'F') d = h - 'A' + 10; else fail("bad unicode escape");
              cp = cp * 16 + (uint32_t)d;
            }
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: fail("bad escape");
        }
      } else out += c;
    }
    return out;
  }
  std::string parse_literal_string() {
    if (peek() == '\'' && peek(1) == '\'' && peek(2) == '\'') fail("multi-line strings are not supported");
    next();
    std::string out;
    while (true) {
      if (eof() || peek() == '\n') fail("unterminated string");
      char c = next();
      if (c == '\'') break;
      out += c;
    }
    return out;
  }
  std::string parse_key_part() {
    skip_ws();
    if (peek() == '"') return parse_basic_string();
    if (peek() == '\'') return parse_literal_string();
    std::string k;
    while (!eof() && bare_char(peek())) k += next();
    if (k.empty()) fail("expected key");
    return k;
  }
  std::vector<std::string> parse_dotted_key() {
    std::vector<std::string> parts;
    parts.push_back(parse_key_part());
    skip_ws();
    while (peek() == '.') { next(); parts.push_back(parse_key_part()); skip_ws(); }
    return parts;
  }

  // descend into `k` of table `t`, creating a table if absent; arrays of tables resolve to their last element
  Value* descend(Value* t, const std::string& k) {
    auto it = t->tab.find(k);
    if (it == t->tab.end()) {
      auto nv = std::make_shared<Value>(); nv->kind = Value::TABLE;
      t->tab[k] = nv; t->key_order.push_back(k);
      return nv.get();
    }
    Value* v = it->second.get();
    if (v->kind == Value::TABLE) { if (v->inline_closed) fail("cannot extend inline table `" + k + "`"); return v; }
    if (v->kind == Value::ARRAY && v->is_table_array && !v->arr.empty()) return v->arr.back().get();
    fail("key `" + k + "` is not a table");
  }

  Value* parse_header() {
    next();  // '['
    bool is_array = false;
    if (peek() == '[') { next(); is_array = true; }
    std::vector<std::string> parts = parse_dotted_key();
    skip_ws();
    if (next() != ']') fail("expected ]");
    if (is_array && next() != ']') fail("expected ]]");
    Value* t = root_.get();
    for (size_t i = 0; i + 1 < parts.size(); ++i) t = descend(t, parts[i]);
    const std::string& last = parts.back();
    if (!is_array) return descend(t, last);
    auto it = t->tab.find(last);
    Value* arr;
    if (it == t->tab.end()) {
      auto nv = std::make_shared<Value>(); nv->kind = Value::ARRAY; nv->is_table_array = true;
      t->tab[last] = nv; t->key_order.push_back(last);
      arr = nv.get();
    } else {
      arr = it->second.get();
      if (arr->kind != Value::ARRAY || !arr->is_table_array) fail("key `" + last + "` is not an array of tables");
    }
    auto elem = std::make_shared<Value>(); elem->kind = Value::TABLE;
    arr->arr.push_back(elem);
    return elem.get();
  }

  void parse_keyval(Value* table) {
    std::vector<std::string> parts = parse_dotted_key();
    skip_ws();
    if (eof() || next() != '=') fail("expected =");
    skip_ws();
    ValuePtr v = parse_value();
    Value* t = table;
    for (size_t i = 0; i + 1 < parts.size(); ++i) t = descend(t, parts[i]);
    if (t->tab.count(parts.back())) fail("duplicate key `" + parts.back() + "`");
    t->tab[parts.back()] = v; t->key_order.push_back(parts.back());
  }

  static constexpr int kMaxDepth = 64;
  struct DepthGuard { int& d; explicit DepthGuard(int& x) : d(x) { ++d; } ~DepthGuard() { --d; } };
  ValuePtr parse_value() {
    DepthGuard guard(depth_);
    if (depth_ > kMaxDepth) fail("arrays / inline tables nested deeper than 64");
    auto v = std::make_shared<Value>();
    char c = peek();
    if (c == '"') { v->kind = Value::STRING; v->s = parse_basic_string(); return v; }
    if (c == '\'') { v->kind = Value::STRING; v->s = parse_literal_string(); return v; }
    if (c == '[') {
      next();
      v->kind = Value::ARRAY; v->inline_closed = true;
      while (true) {
        skip_ws_nl();
        if (peek() == ']') { next(); break; }
        v->arr.push_back(parse_value());
        skip_ws_nl();
        if (peek() == ',') { next(); continue; }
        skip_ws_nl();
        if (peek() == ']') { next(); break; }
        fail("expected , or ] in array");
      }
      return v;
    }
    if (c == '{') {
      next();
      v->kind = Value::TABLE;
      skip_ws();
      if (peek() == '}') { next(); v->inline_closed = true; return v; }
      while (true) {
        skip_ws();
        parse_keyval(v.get());
        skip_ws();
        if (peek() == ',') { next(); continue; }
        if (peek() == '}') { next(); break; }
        fail("expected , or } in inline table");
      }
      v->inline_closed = true;
      return v;
    }
    if (t_.compare(p_, 4, "true") == 0 && !bare_char(peek(4))) { p_ += 4; v->kind = Value::BOOL; v->b = true; return v; }
    if (t_.compare(p_, 5, "false") == 0 && !bare_char(peek(5))) { p_ += 5; v->kind = Value::BOOL; v->b = false; return v; }
    // number
    std::string tok;
    while (!eof()) {
      char d = peek();
      if ((d >= '0' && d <= '9') || d == '+' || d == '-' || d == '.' || d == 'e' || d == 'E' || d == '_' ||
          d == 'i' || d == 'n' || d == 'f' || d == 'a' || d == 'x' || d == 'o' || d == 'b' ||
          (d >= 'A' && d <= 'F') || (d >= 'a' && d <= 'f') || d == ':' || d == 'T' || d == 'Z')
        tok += next();
      else break;
    }
    if (tok.empty()) fail("expected value");
    if (tok.find(':') != std::string::npos || (tok.size() > 4 && tok[4] == '-' && tok[0] >= '0' && tok[0] <= '9' && tok.find('e') == std::string::npos && tok.find('.') == std::string::npos && tok.find('-', 5) != std::string::npos))
      fail("date/time values are not supported");
    std::string clean;
    for (char d : tok) if (d != '_') clean += d;
    std::string body = clean;
    if (!body.empty() && (body[0] == '+' || body[0] == '-')) body = body.substr(1);
    if (body == "inf" || body == "nan") {
      v->kind = Value::FLOAT;
      v->f = body == "inf" ? (clean[0] == '-' ? -HUGE_VAL : HUGE_VAL) : std::strtod("nan", nullptr);
      return v;
    }
    bool is_float = false;
    if (body.size() > 2 && body[0] == '0' && (body[1] == 'x' || body[1] == 'o' || body[1] == 'b')) {
      int base = body[1] == 'x' ? 16 : (body[1] == 'o' ? 8 : 2);
      char* end = nullptr;
      v->kind = Value::INT; v->i = (int64_t)std::strtoll(body.c_str() + 2, &end, base);
      if (*end) fail("bad integer `" + tok + "`");
      return v;
    }
    for (char d : body) {
      if (d == '.' || d == 'e' || d == 'E') is_float = true;
      else if (!((d >= '0' && d <= '9') || d == '+' || d == '-')) fail("bad number `" + tok + "`");
    }
    char* end = nullptr;
    if (is_float) { v->kind = Value::FLOAT; v->f = std::strtod(clean.c_str(), &end); }
    else { v->kind = Value::INT; v->i = (int64_t)std::strtoll(clean.c_str(), &end, 10); }
    if (!end || *end) fail("bad number `" + tok + "`");
    return v;
  }
};

inline ValuePtr parse(const std::string& text) { Parser p(text); return p.parse(); }

}  // namespace lrtoml
