// Minimal XML subset reader: elements, attributes, comments, <?...?> and <!...> declarations.
// Character data is skipped (the reference's scene dialect carries everything in attributes,
// reference RayTracer/xmlload.h:57-69).  Stands in for tinyxml2, whose header is absent from the
// reference checkout.
#pragma once
#include <cctype>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace rth {

struct XmlElem {
  std::string tag;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<std::unique_ptr<XmlElem>> children;

  const char* attr(const char* name) const {
    for (auto& a : attrs)
      if (a.first == name) return a.second.c_str();
    return nullptr;
  }
  static bool ieq(const char* a, const char* b) {
    if (!a || !b) return a == b;
    for (;; ++a, ++b) {
      if (std::tolower((unsigned char)*a) != std::tolower((unsigned char)*b)) return false;
      if (!*a) return true;
    }
  }
  // first child with the given tag (case-sensitive like tinyxml2's FirstChildElement(name))
  const XmlElem* child(const char* name) const {
    for (auto& c : children)
      if (c->tag == name) return c.get();
    return nullptr;
  }
};

class XmlParser {
 public:
  // Returns the synthetic document node holding the top-level elements, or nullptr on error.
  std::unique_ptr<XmlElem> parse(const std::string& text, std::string& err) {
    s_ = text.c_str();
    n_ = text.size();
    i_ = 0;
    auto doc = std::make_unique<XmlElem>();
    doc->tag = "#document";
    while (true) {
      skip_misc();
      if (i_ >= n_) break;
      if (s_[i_] != '<') { ++i_; continue; }
      auto e = parse_elem(err);
      if (!e) return nullptr;
      doc->children.push_back(std::move(e));
    }
    return doc;
  }

 private:
  const char* s_ = nullptr;
  size_t n_ = 0, i_ = 0;

  bool starts(const char* p) const {
    size_t k = 0;
    while (p[k]) {
      if (i_ + k >= n_ || s_[i_ + k] != p[k]) return false;
      ++k;
    }
    return true;
  }
  void skip_ws() {
    while (i_ < n_ && std::isspace((unsigned char)s_[i_])) ++i_;
  }
  // skips whitespace, text, comments, declarations; stops at an element start or end tag
  void skip_misc() {
    while (i_ < n_) {
      if (s_[i_] != '<') { ++i_; continue; }
      if (starts("<!--")) {
        size_t e = n_;
        for (size_t k = i_ + 4; k + 3 <= n_; ++k)
          if (s_[k] == '-' && s_[k + 1] == '-' && s_[k + 2] == '>') { e = k + 3; break; }
        i_ = e;
        continue;
      }
      if (starts("<?") || starts("<!")) {
        while (i_ < n_ && s_[i_] != '>') ++i_;
        if (i_ < n_) ++i_;
        continue;
      }
      return;
    }
  }
  static bool name_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }
  std::string parse_name() {
    size_t b = i_;
    while (i_ < n_ && name_char(s_[i_])) ++i_;
    return std::string(s_ + b, i_ - b);
  }
  static std::string unescape(const std::string& v) {
    if (v.find('&') == std::string::npos) return v;
    std::string o;
    for (size_t k = 0; k < v.size(); ++k) {
      if (v[k] == '&') {
        if (!v.compare(k, 5, "&amp;")) { o += '&'; k += 4; continue; }
        if (!v.compare(k, 4, "&lt;")) { o += '<'; k += 3; continue; }
        if (!v.compare(k, 4, "&gt;")) { o += '>'; k += 3; continue; }
        if (!v.compare(k, 6, "&quot;")) { o += '"'; k += 5; continue; }
        if (!v.compare(k, 6, "&apos;")) { o += '\''; k += 5; continue; }
      }
      o += v[k];
    }
    return o;
  }
  std::unique_ptr<XmlElem> parse_elem(std::string& err) {
    // at '<'
    ++i_;
    auto e = std::make_unique<XmlElem>();
    e->tag = parse_name();
    if (e->tag.empty()) { err = "xml: empty tag name at offset " + std::to_string(i_); return nullptr; }
    while (true) {
      skip_ws();
      if (i_ >= n_) { err = "xml: unexpected end inside <" + e->tag + ">"; return nullptr; }
      if (s_[i_] == '/') {
        if (i_ + 1 < n_ && s_[i_ + 1] == '>') { i_ += 2; return e; }
        err = "xml: stray '/' in <" + e->tag + ">";
        return nullptr;
      }
      if (s_[i_] == '>') { ++i_; break; }
      std::string an = parse_name();
      if (an.empty()) { err = "xml: bad attribute in <" + e->tag + ">"; return nullptr; }
      skip_ws();
      if (i_ >= n_ || s_[i_] != '=') { err = "xml: attribute '" + an + "' without value"; return nullptr; }
      ++i_;
      skip_ws();
      if (i_ >= n_ || (s_[i_] != '"' && s_[i_] != '\'')) { err = "xml: unquoted attribute '" + an + "'"; return nullptr; }
      char q = s_[i_++];
      size_t b = i_;
      while (i_ < n_ && s_[i_] != q) ++i_;
      if (i_ >= n_) { err = "xml: unterminated attribute '" + an + "'"; return nullptr; }
      e->attrs.emplace_back(an, unescape(std::string(s_ + b, i_ - b)));
      ++i_;
    }
    // children until </tag>
    while (true) {
      skip_misc();
      if (i_ >= n_) { err = "xml: missing </" + e->tag + ">"; return nullptr; }
      if (starts("</")) {
        i_ += 2;
        std::string cn = parse_name();
        skip_ws();
        if (i_ < n_ && s_[i_] == '>') ++i_;
        if (cn != e->tag) { err = "xml: </" + cn + "> closes <" + e->tag + ">"; return nullptr; }
        return e;
      }
      auto c = parse_elem(err);
      if (!c) return nullptr;
      e->children.push_back(std::move(c));
    }
  }
};

}  // namespace rth
