#include "ini_config.h"

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace rgpu_host {

namespace {
std::string rstrip(std::string s) {
  while (!s.empty() && std::isspace(static_cast<unsigned char>(s.back()))) s.pop_back();
  return s;
}
std::string lskip(const std::string& s) {
  size_t i = 0;
  while (i < s.size() && std::isspace(static_cast<unsigned char>(s[i]))) ++i;
  return s.substr(i);
}
// position of the first `c`, or of a ';' that follows a whitespace, or npos-as-size (ini.cpp:45-53)
size_t find_char_or_comment(const std::string& s, size_t from, char c) {
  bool was_ws = false;
  size_t i = from;
  while (i < s.size() && s[i] != c && !(was_ws && s[i] == ';')) {
    was_ws = std::isspace(static_cast<unsigned char>(s[i])) != 0;
    ++i;
  }
  return i;
}
}  // namespace

std::string IniConfig::make_key(const std::string& section, const std::string& name) {
  std::string key = section + "." + name;
  for (size_t i = 0; i < key.size(); ++i) key[i] = static_cast<char>(std::tolower(static_cast<unsigned char>(key[i])));
  return key;
}

int IniConfig::load_file(const std::string& path) {
  std::ifstream f(path.c_str());
  if (!f) return -1;
  std::stringstream ss;
  ss << f.rdbuf();
  return load_text(ss.str());
}

int IniConfig::load_text(const std::string& text) {
  std::istringstream in(text);
  std::string raw, section, prev_name;
  int lineno = 0, error = 0;
  while (std::getline(in, raw)) {
    ++lineno;
    const std::string line = rstrip(raw);
    const std::string start = lskip(line);
    const bool indented = start.size() < line.size();
    if (!prev_name.empty() && !start.empty() && indented) {
      values_[make_key(section, prev_name)] = start;  // continuation line replaces the value
    } else if (start.empty() || start[0] == ';' || start[0] == '#') {
      // blank or comment
    } else if (start[0] == '[') {
      const size_t end = find_char_or_comment(start, 1, ']');
      if (end < start.size() && start[end] == ']') {
        section = start.substr(1, end - 1);
        prev_name.clear();
      } else if (!error) {
        error = lineno;
      }
    } else {
      const size_t eq = find_char_or_comment(start, 0, '=');
      if (eq < start.size() && start[eq] == '=') {
        const std::string name = rstrip(start.substr(0, eq));
        std::string value = lskip(start.substr(eq + 1));
        const size_t end = find_char_or_comment(value, 0, '\0');
        if (end < value.size() && value[end] == ';') value = value.substr(0, end);
        value = rstrip(value);
        prev_name = name;
        values_[make_key(section, name)] = value;
      } else if (!error) {
        error = lineno;
      }
    }
  }
  return error;
}

void IniConfig::apply_overrides(const std::string& overrides) {
  size_t pos = 0;
  while (pos < overrides.size()) {
    size_t end = overrides.find(';', pos);
    if (end == std::string::npos) end = overrides.size();
    const std::string item = overrides.substr(pos, end - pos);
    pos = end + 1;
    const size_t eq = item.find('=');
    const size_t dot = item.find('.');
    if (eq == std::string::npos || dot == std::string::npos || dot > eq) continue;
    const std::string section = rstrip(lskip(item.substr(0, dot)));
    const std::string name = rstrip(lskip(item.substr(dot + 1, eq - dot - 1)));
    const std::string value = rstrip(lskip(item.substr(eq + 1)));
    values_[make_key(section, name)] = value;
  }
}

std::string IniConfig::get_string(const std::string& section, const std::string& name, const std::string& dflt) const {
  std::map<std::string, std::string>::const_iterator it = values_.find(make_key(section, name));
  return it == values_.end() ? dflt : it->second;
}

long IniConfig::get_integer(const std::string& section, const std::string& name, long dflt) const {
  const std::string v = get_string(section, name, "");
  const char* s = v.c_str();
  char* end = 0;
  const long n = std::strtol(s, &end, 0);
  return end > s ? n : dflt;
}

float IniConfig::get_float(const std::string& section, const std::string& name, float dflt) const {
  const std::string v = get_string(section, name, "");
  const char* s = v.c_str();
  char* end = 0;
  const float x = std::strtof(s, &end);
  return end > s ? x : dflt;
}

bool IniConfig::get_bool(const std::string& section, const std::string& name, bool dflt) const {
  const std::string v = get_string(section, name, "");
  bool val = dflt;
  if (v == "1" || v == "yes" || v == "true" || v == "on") val = true;
  if (v == "0" || v == "no" || v == "false" || v == "off") val = false;
  if (v.empty()) val = dflt;
  return val;
}

void IniConfig::set_string(const std::string& section, const std::string& name, const std::string& value) {
  values_[make_key(section, name)] = value;
}

}  // namespace rgpu_host
