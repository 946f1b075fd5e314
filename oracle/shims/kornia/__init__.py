"""TEST INFRASTRUCTURE ONLY - see kornia/utils/__init__.py."""
