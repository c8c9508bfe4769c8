// needle — command-line front-end over libneedle_capi.so with the reference CLI's surface
// (needle/src/main.rs:13-339): subcommands info / analyze / search, the same flags, defaults and validation.
// It calls nothing but the C ABI of include/needle.h (plus two diagnostics of needle_hip.h for `info`), i.e.
// exactly what a program linked against the reference's needle-capi would call.
//
// Differences that follow from this build's scope: media files are RIFF/WAVE PCM (no FFmpeg), so `info`
// reports the library and device instead of an FFmpeg version, and --threaded-decoding is accepted and ignored.
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/needle_hip.h"

namespace {

const char *kUsage =
    "USAGE:\n"
    "    needle [--no-threading] [--file-headers-only] <SUBCOMMAND>\n\n"
    "SUBCOMMANDS:\n"
    "    info       Displays info about needle and its dependencies.\n"
    "    analyze    <PATHS>... [-m|--mode audio] [--opening-search-percentage F] [--ending-search-percentage F]\n"
    "               [--hash-duration F] [--include-endings] [--threaded-decoding] [--force]\n"
    "    search     <PATHS>... [--hash-match-threshold N] [--min-opening-duration N] [--min-ending-duration N]\n"
    "               [--time-padding F] [--analyze] [--use-skip-files] [--write-skip-files] [--include-endings]\n"
    "               [--no-display]\n";

// clap's `cmd.error(kind, msg).exit()`: message + usage on stderr, exit status 2 (main.rs:196-251)
[[noreturn]] void usage_error(const std::string &msg) {
  std::fprintf(stderr, "error: %s\n\n%s\nFor more information try --help\n", msg.c_str(), kUsage);
  std::exit(2);
}

float parse_f32(const std::string &flag, const char *v) {
  errno = 0;
  char *end = nullptr;
  const float x = std::strtof(v, &end);
  if (errno != 0 || end == v || *end != '\0') usage_error("Invalid value \"" + std::string(v) + "\" for '--" + flag + "'");
  return x;
}

uint16_t parse_u16(const std::string &flag, const char *v) {
  errno = 0;
  char *end = nullptr;
  const unsigned long x = std::strtoul(v, &end, 10);
  if (errno != 0 || end == v || *end != '\0' || x > 65535 || v[0] == '-')
    usage_error("Invalid value \"" + std::string(v) + "\" for '--" + flag + "'");
  return (uint16_t)x;
}

struct Args {
  std::string command;
  bool no_threading = false, file_headers_only = false;
  std::vector<std::string> paths;
  // analyze (audio/mod.rs:14-45 defaults)
  float opening_search_percentage = 0.50f, ending_search_percentage = 0.25f, hash_duration = 0.3f;
  bool include_endings = false, threaded_decoding = false, force = false;
  // search
  uint16_t hash_match_threshold = 10, min_opening_duration = 20, min_ending_duration = 20;
  float time_padding = 0.0f;
  bool analyze = false, use_skip_files = false, write_skip_files = false, no_display = false;
};

Args parse(int argc, char **argv) {
  Args a;
  auto value_of = [&](int &i, const std::string &flag, const char *inline_value) -> const char * {
    if (inline_value) return inline_value;
    if (i + 1 >= argc) usage_error("The argument '--" + flag + " <" + flag + ">' requires a value but none was supplied");
    return argv[++i];
  };
  bool only_paths = false;
  for (int i = 1; i < argc; i++) {
    std::string tok = argv[i];
    if (only_paths || tok.empty() || tok[0] != '-' || tok == "-") {
      if (a.command.empty()) {
        if (tok != "info" && tok != "analyze" && tok != "search")
          usage_error("Found argument '" + tok + "' which wasn't expected, or isn't valid in this context");
        a.command = tok;
      } else {
        a.paths.push_back(tok);
      }
      continue;
    }
    if (tok == "--") { only_paths = true; continue; }
    if (tok == "-h" || tok == "--help") { std::printf("needle %s\n\n%s", needle_hip_version(), kUsage); std::exit(0); }
    if (tok == "-V" || tok == "--version") { std::printf("needle %s\n", needle_hip_version()); std::exit(0); }
    const char *inline_value = nullptr;
    std::string flag;
    if (tok.rfind("--", 0) == 0) {
      const size_t eq = tok.find('=');
      flag = tok.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
      if (eq != std::string::npos) inline_value = argv[i] + eq + 1;
    } else if (tok == "-m") {
      flag = "mode";
    } else {
      usage_error("Found argument '" + tok + "' which wasn't expected, or isn't valid in this context");
    }
    // global flags are valid anywhere (clap `global = true`, main.rs:176-192)
    if (flag == "no-threading") { a.no_threading = true; continue; }
    if (flag == "file-headers-only") { a.file_headers_only = true; continue; }
    const bool an = a.command == "analyze", se = a.command == "search";
    if (an && flag == "mode") {
      const std::string m = value_of(i, flag, inline_value);
      if (m != "audio") usage_error("\"" + m + "\" isn't a valid value for '--mode <MODE>'\n\t[possible values: audio]");
    } else if (an && flag == "opening-search-percentage") a.opening_search_percentage = parse_f32(flag, value_of(i, flag, inline_value));
    else if (an && flag == "ending-search-percentage") a.ending_search_percentage = parse_f32(flag, value_of(i, flag, inline_value));
    else if (an && flag == "hash-duration") a.hash_duration = parse_f32(flag, value_of(i, flag, inline_value));
    else if ((an || se) && flag == "include-endings") a.include_endings = true;
    else if (an && flag == "threaded-decoding") a.threaded_decoding = true;
    else if (an && flag == "force") a.force = true;
    else if (se && flag == "hash-match-threshold") a.hash_match_threshold = parse_u16(flag, value_of(i, flag, inline_value));
    else if (se && flag == "min-opening-duration") a.min_opening_duration = parse_u16(flag, value_of(i, flag, inline_value));
    else if (se && flag == "min-ending-duration") a.min_ending_duration = parse_u16(flag, value_of(i, flag, inline_value));
    else if (se && flag == "time-padding") a.time_padding = parse_f32(flag, value_of(i, flag, inline_value));
    else if (se && flag == "analyze") a.analyze = true;
    else if (se && flag == "use-skip-files") a.use_skip_files = true;
    else if (se && flag == "write-skip-files") a.write_skip_files = true;
    else if (se && flag == "no-display") a.no_display = true;
    else usage_error("Found argument '" + tok + "' which wasn't expected, or isn't valid in this context");
  }
  if (a.command.empty()) usage_error("'needle' requires a subcommand, but one was not provided");
  // Cli::validate, main.rs:196-241
  if (a.command == "analyze") {
    if (a.paths.empty()) usage_error("The following required arguments were not provided:\n    <PATHS>...");
    if (a.opening_search_percentage >= 1.0f) usage_error("opening_search_percentage must be less than 1.0");
    if (a.ending_search_percentage >= 1.0f) usage_error("ending_search_percentage must be less than 1.0");
    if (a.hash_duration <= 0.0f) usage_error("hash_duration must be greater than 0");
  } else if (a.command == "search") {
    if (a.paths.empty()) usage_error("The following required arguments were not provided:\n    <PATHS>...");
    if (a.hash_match_threshold > 32) usage_error("hash_match_threshold cannot be larger than 32");
  }
  return a;
}

// Cli::find_video_files (main.rs:243-251) + videos.sort() (main.rs:272,305)
std::vector<std::string> find_videos(const Args &a) {
  std::vector<const char *> raw;
  for (const std::string &p : a.paths) raw.push_back(p.c_str());
  const char *const *videos = nullptr;
  size_t n = 0;
  const NeedleError e = needle_util_find_video_files(raw.data(), raw.size(), !a.file_headers_only, true, &videos, &n);
  if (e != NeedleError_Ok) {
    const char *detail = needle_hip_last_error_message();
    usage_error(detail && *detail ? detail : needle_error_to_str(e));
  }
  std::vector<std::string> out;
  for (size_t i = 0; i < n; i++) out.push_back(videos[i]);
  needle_util_video_files_free(videos, n);
  std::sort(out.begin(), out.end());
  return out;
}

int fail(NeedleError e) {  // `fn main() -> needle::Result<()>`: Err -> "Error: ..." on stderr, status 1
  const char *detail = needle_hip_last_error_message();
  std::fprintf(stderr, "Error: %s%s%s\n", needle_error_to_str(e), detail && *detail ? ": " : "", detail ? detail : "");
  return 1;
}

}  // namespace

int main(int argc, char **argv) {
  const Args a = parse(argc, argv);
  if (a.command == "info") {
    int devices = 0;
    needle_hip_device_count(&devices);
    std::printf("needle version: %s\n", needle_hip_version());
    std::printf("FFmpeg version: none (this build reads RIFF/WAVE PCM; any sample rate, mono or stereo)\n");
    std::printf("HIP devices: %d\n", devices);
    return 0;
  }
  const std::vector<std::string> videos = find_videos(a);
  std::vector<const char *> raw;
  for (const std::string &v : videos) raw.push_back(v.c_str());
  if (a.command == "analyze") {
    if (videos.empty()) return fail(NeedleError_InvalidArgument);  // Error::AnalyzerMissingPaths (analyzer.rs:431-433)
    NeedleAudioAnalyzer *analyzer = nullptr;
    NeedleError e = needle_audio_analyzer_new(raw.data(), raw.size(), a.opening_search_percentage,
                                              a.ending_search_percentage, a.include_endings, a.threaded_decoding,
                                              a.force, &analyzer);
    if (e != NeedleError_Ok) return fail(e);
    e = needle_audio_analyzer_run(analyzer, a.hash_duration, /*persist=*/true, !a.no_threading);  // main.rs:288
    needle_audio_analyzer_free(analyzer);
    return e == NeedleError_Ok ? 0 : fail(e);
  }
  // search (main.rs:291-332)
  if (videos.size() < 2)
    usage_error("need at least 2 valid video files, but only found " + std::to_string(a.paths.size()) +
                " in provided video paths");
  const NeedleAudioComparator *comparator = nullptr;
  NeedleError e = needle_audio_comparator_new(raw.data(), raw.size(), a.include_endings, a.hash_match_threshold,
                                              a.min_opening_duration, a.min_ending_duration, a.time_padding,
                                              &comparator);
  if (e != NeedleError_Ok) return fail(e);
  e = needle_audio_comparator_run(comparator, a.analyze, !a.no_display, a.use_skip_files, a.write_skip_files,
                                  !a.no_threading);
  needle_audio_comparator_free(comparator);
  return e == NeedleError_Ok ? 0 : fail(e);
}
