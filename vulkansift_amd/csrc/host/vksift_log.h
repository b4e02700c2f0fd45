/* vksift_log.h — 4-level logger behind vksift_setLogLevel (reference: src/vulkansift/vkenv/logger.{c,h}). */
#ifndef VKSIFT_LOG_H
#define VKSIFT_LOG_H

typedef enum
{
  VKSIFT_LOGLVL_NONE = 0,
  VKSIFT_LOGLVL_ERROR,
  VKSIFT_LOGLVL_WARNING,
  VKSIFT_LOGLVL_INFO,
  VKSIFT_LOGLVL_DEBUG
} vksift_log_level;

void vksift_log_set_level(vksift_log_level lvl);
vksift_log_level vksift_log_get_level(void);
void vksift_log(vksift_log_level lvl, const char *tag, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

#define logError(tag, ...) vksift_log(VKSIFT_LOGLVL_ERROR, tag, __VA_ARGS__)
#define logWarning(tag, ...) vksift_log(VKSIFT_LOGLVL_WARNING, tag, __VA_ARGS__)
#define logInfo(tag, ...) vksift_log(VKSIFT_LOGLVL_INFO, tag, __VA_ARGS__)
#define logDebug(tag, ...) vksift_log(VKSIFT_LOGLVL_DEBUG, tag, __VA_ARGS__)

#endif
