// ITK-lite: itkTestMain.h -- the test dispatcher the reference driver is built with
// (Testing/CuberilleTest01.cxx:31-33,50-55; CTest invokes `CuberilleTest01 Test01 <args>`,
// Testing/CMakeLists.txt:12-13): main() looks argv[1] up among the registered tests and calls
// it with (argc-1, argv+1).
#ifndef ITK_LITE_TEST_MAIN_H
#define ITK_LITE_TEST_MAIN_H
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>

typedef int (*MainFuncPointer)(int, char *[]);
std::map<std::string, MainFuncPointer> StringToTestFunctionMap;

#define REGISTER_TEST(test)          \
  extern int test(int, char *[]);    \
  StringToTestFunctionMap[#test] = test

void RegisterTests();

int main(int ac, char *av[]) {
  RegisterTests();
  if (ac < 2) {
    std::cout << "Available tests:\n";
    for (std::map<std::string, MainFuncPointer>::iterator j = StringToTestFunctionMap.begin(); j != StringToTestFunctionMap.end(); ++j)
      std::cout << "  " << j->first << "\n";
    return EXIT_FAILURE;
  }
  std::map<std::string, MainFuncPointer>::iterator j = StringToTestFunctionMap.find(av[1]);
  if (j == StringToTestFunctionMap.end()) {
    std::cerr << "Test '" << av[1] << "' is not registered\n";
    return EXIT_FAILURE;
  }
  try {
    return (*j->second)(ac - 1, av + 1);
  } catch (const std::exception &e) {
    std::cerr << "Caught an exception: " << e.what() << std::endl;
  } catch (...) {
    std::cerr << "Caught an unknown exception" << std::endl;
  }
  return EXIT_FAILURE;
}
#endif
