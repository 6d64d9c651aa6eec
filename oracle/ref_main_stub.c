/* oracle/ref_main_stub.c -- TEST INFRASTRUCTURE.  The reference's main.c is compiled into
 * _ref/libbwa_ref.so with -Dmain=bwa_ref_main (so the library also carries main.c's globals,
 * e.g. bwa_pg, and can be dlopen()ed by the Python checkers); this stub turns the library back
 * into the `bwa` command line. */
int bwa_ref_main(int argc, char *argv[]);
int main(int argc, char *argv[]) { return bwa_ref_main(argc, argv); }
