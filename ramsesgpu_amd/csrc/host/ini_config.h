// ini_config.h -- key/value view of a ramsesGPU parameter file.
//
// Host-side mirror of the reference's ConfigMap (src/utils/config/ConfigMap.{h,cpp}) on top of its INIReader /
// inih parser (src/utils/config/inih/INIReader.cpp, ini.cpp).  Same observable semantics, own implementation:
//   * keys are "section.name", lower-cased (INIReader.cpp:94-101)  -> lookups are case-insensitive
//   * '#' or ';' at line start is a comment; " ;" (semicolon after whitespace) starts an inline comment
//     (ini.cpp:45-53,102-128); a non-blank line starting with whitespace continues -- and REPLACES -- the previous
//     name's value (ini.cpp:93-100 + INIReader.cpp:107-112)
//   * getInteger = strtol(base 0) (INIReader.cpp:61-69); getFloat = strtof, i.e. every real knob is parsed as
//     a FLOAT and then widened (ConfigMap.cpp:41-49) -- parity-critical, e.g. gamma0=1.666 -> 1.66600000858...
//   * getBool accepts 1/yes/true/on and 0/no/false/off, anything else -> default (ConfigMap.cpp:65-86)
#pragma once
#include <map>
#include <string>

namespace rgpu_host {

class IniConfig {
 public:
  IniConfig() {}
  // returns 0, -1 if the file cannot be opened, or the first line number in error
  int load_file(const std::string& path);
  int load_text(const std::string& text);
  // "section.key=value;section.key=value" (bench / test overrides; not a reference feature)
  void apply_overrides(const std::string& overrides);

  std::string get_string(const std::string& section, const std::string& name, const std::string& dflt) const;
  long get_integer(const std::string& section, const std::string& name, long dflt) const;
  float get_float(const std::string& section, const std::string& name, float dflt) const;
  bool get_bool(const std::string& section, const std::string& name, bool dflt) const;
  void set_string(const std::string& section, const std::string& name, const std::string& value);
  const std::map<std::string, std::string>& values() const { return values_; }

 private:
  static std::string make_key(const std::string& section, const std::string& name);
  std::map<std::string, std::string> values_;
};

}  // namespace rgpu_host
