/*
 * jni.h -- a STAND-IN for the JDK's header, for type-checking bindings/fmx_jni.c on machines without a JDK
 * (tests/test_jni_shim.py runs `gcc -std=c99 -fsyntax-only -Wall -Werror` over the shim with this directory on the
 * include path).  It declares, from the JNI specification's signatures, exactly the types and the JNIEnv
 * functions the shim uses.  It is NOT the real function table: the members are in alphabetical order, not in the
 * specification's slot order, so nothing compiled against it can be loaded into a JVM.  A build for a JVM uses the
 * JDK's own jni.h.
 */
#ifndef FMX_TEST_JNI_STANDIN_H
#define FMX_TEST_JNI_STANDIN_H

#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

typedef uint8_t jboolean;
typedef int8_t jbyte;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

struct fmx_standin_jobject;
typedef struct fmx_standin_jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jbyteArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv *env, const char *name);
  jsize (*GetArrayLength)(JNIEnv *env, jarray array);
  void (*GetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, jbyte *buf);
  void *(*GetDirectBufferAddress)(JNIEnv *env, jobject buf);
  jlong (*GetDirectBufferCapacity)(JNIEnv *env, jobject buf);
  void (*GetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, jint *buf);
  void (*GetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, jlong *buf);
  void *(*GetPrimitiveArrayCritical)(JNIEnv *env, jarray array, jboolean *isCopy);
  const char *(*GetStringUTFChars)(JNIEnv *env, jstring str, jboolean *isCopy);
  jbyteArray (*NewByteArray)(JNIEnv *env, jsize len);
  jobject (*NewDirectByteBuffer)(JNIEnv *env, void *address, jlong capacity);
  void (*ReleasePrimitiveArrayCritical)(JNIEnv *env, jarray array, void *carray, jint mode);
  void (*ReleaseStringUTFChars)(JNIEnv *env, jstring str, const char *chars);
  void (*SetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize len, const jbyte *buf);
  void (*SetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize len, const jdouble *buf);
  void (*SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize len, const jint *buf);
  void (*SetLongArrayRegion)(JNIEnv *env, jlongArray array, jsize start, jsize len, const jlong *buf);
  jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
};

#endif
