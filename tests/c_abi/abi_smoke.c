/* A plain C consumer of include/needle.h: constructs, inspects and frees the opaque handles the way a
 * needle-capi client does (create -> print paths -> run -> free, errors through needle_error_to_str).
 * Runs without a GPU: the run calls are expected to fail cleanly on paths that do not exist. */
#include <stdio.h>
#include <string.h>

#include <needle.h>

int main(void) {
  NeedleError err;
  NeedleAudioAnalyzer *analyzer = NULL;
  const NeedleAudioComparator *comparator = NULL;
  const FrameHashes *frame_hashes = NULL;
  const char *paths[] = {"/tmp/needle-abi-smoke-a.wav", "/tmp/needle-abi-smoke-b.wav"};

  err = needle_audio_analyzer_new(paths, 2, 0.5f, 0.25f, false, false, true, &analyzer);
  if (err != NeedleError_Ok || analyzer == NULL) return 1;
  needle_audio_analyzer_print_paths(analyzer);
  err = needle_audio_analyzer_run(analyzer, 0.3f, false, true);
  if (err == NeedleError_Ok) return 2; /* the files do not exist */
  printf("analyzer run on missing files: %s\n", needle_error_to_str(err));
  if (needle_audio_analyzer_get_frame_hashes(analyzer, 0, &frame_hashes) != NeedleError_InvalidArgument) return 3;

  err = needle_audio_comparator_new(paths, 2, true, 10, 20, 20, 0.0f, &comparator);
  if (err != NeedleError_Ok || comparator == NULL) return 4;
  err = needle_audio_comparator_run(comparator, false, false, false, false, true);
  if (err != NeedleError_FrameHashDataNotFound) return 5;
  if (strcmp(needle_error_to_str(err), "Frame hash data not found on disk") != 0) return 6;
  if (needle_audio_comparator_new_default(paths, 1, &comparator) != NeedleError_ComparatorMinimumPaths) return 7;

  needle_audio_analyzer_free(analyzer);
  needle_audio_comparator_free(comparator);
  needle_util_video_files_free(NULL, 0);
  printf("abi smoke ok\n");
  return 0;
}
